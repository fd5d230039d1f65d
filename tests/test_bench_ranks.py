"""CPU, world_size = 2, gloo: bench.py's OWN multi-rank code -- ``open_world`` (process group), ``build_attacker`` (image
shards / one surrogate per rank through ``dist.sharded_attack``), ``timed_region`` (warm-up, barrier, K steps, barrier),
``over_ranks`` (MAX all-reduce of the time, all-gather of the rates) and the JSON line with ``n_gpus`` / ``ranks_observed`` /
``collective_backend`` -- before the driver's 8-GPU node is the first to run it.  The kernels run from their host build
(tests/host_kernels.py) on CPU tensors; only the backend differs from the GPU launch (gloo instead of RCCL)."""
import io
import json
import os
import socket
import sys
import contextlib

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)

    def setenv(self, name, value):
        os.environ[name] = value


def _rank(rank, world, port, out, argv):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import host_kernels
    host_kernels.install(_Patch())
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(argv)
    with open("%s.%d" % (out, rank), "w") as fh:
        fh.write(buf.getvalue())


def _bench(tmp_path, argv, world=2):
    out = str(tmp_path / "line")
    mp.spawn(_rank, args=(world, _free_port(), out, argv), nprocs=world, join=True)
    lines = [open("%s.%d" % (out, r)).read().strip() for r in range(world)]
    assert all(not text for text in lines[1:]), "only rank 0 prints"
    assert len(lines[0].splitlines()) == 1, "exactly one JSON line"
    return json.loads(lines[0])


COMMON = ["--steps", "2", "--warmup", "1", "--batch", "2", "--image-size", "32", "--classes", "10", "--backend", "gloo",
          "--device", "cpu", "--cpu-images", "0", "--kernel-sweep", "0", "--fold-bn", "0", "--channels-last", "0"]


def test_bench_two_ranks_image_shards(tmp_path):
    """bench.py --gpus 2: two ranks, each its own batches (no data-path collective); value = images of both ranks over
    the MAX time, per-rank rates gathered"""
    r = _bench(tmp_path, ["--gpus", "2", "--attack", "mifgsm", "--model", "toy_cnn"] + COMMON)
    print(json.dumps(r))
    assert r["n_gpus"] == 2 and r["config"]["ranks_observed"] == 2 and r["config"]["collective_backend"] == "gloo"
    assert r["config"]["gpus_requested"] == 2 and len(r["config"]["images_per_s_per_rank"]) == 2
    assert r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    # whole-job rate: 2 ranks x 2 steps x 2 images over the slowest rank's bracketed time
    assert abs(r["value"] - 2 * 2 * 2 / (r["ms_per_step"] * 2 / 1e3)) <= 1e-2 * r["value"]
    assert r["value"] <= sum(r["config"]["images_per_s_per_rank"]) * 1.001
    assert "image-shard x2" in r["config"]["parallelism"]
    # rank 0 warmed up first, rank 1 after it (bench.staged_warmup): the time before the timed region holds both warm-ups
    st = r["config"]["startup"]
    assert st["staged"] is True and st["until_timed_region_s"] >= st["rank0_warmup_s"] > 0


def test_bench_two_ranks_sharded_ensemble(tmp_path):
    """bench.py --gpus 2 --attack ens --model a,b: one surrogate per rank through dist.sharded_attack, the two all-reduces
    of the ensemble path in every iteration; one image shard (both ranks hold the same batch)"""
    r = _bench(tmp_path, ["--gpus", "2", "--attack", "ens", "--model", "toy_cnn,toy_cnn"] + COMMON)
    print(json.dumps(r))
    assert r["n_gpus"] == 2 and r["config"]["ranks_observed"] == 2 and r["config"]["collective_backend"] == "gloo"
    assert "1 image shard(s) x 2 model ranks" in r["config"]["parallelism"]
    assert abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 2 / 1e3)) <= 1e-2 * r["value"]       # one shard: 2 steps x 2 images
    assert r["config"]["startup"]["staged"] is False          # the members' steps exchange data: no rank can warm up alone


def test_bench_single_process_same_code(tmp_path):
    """world of one through the same functions (no group): n_gpus 1, backend 'none'"""
    r = _bench(tmp_path, ["--gpus", "1", "--attack", "mifgsm", "--model", "toy_cnn"] + COMMON, world=1)
    assert r["n_gpus"] == 1 and r["config"]["collective_backend"].startswith("none")
