"""GPU (-m gpu): the torch.distributed paths of transferattack_amd/dist.py over RCCL (backend "nccl") on HIP tensors.

The GPU box has ONE device, so the world has one rank: the collectives degenerate, but everything around them is the
real thing -- process-group creation with a device id, sub-groups (model_groups), device-tensor all-reduce / broadcast /
all-gather through RCCL, the autograd Functions of ShardedEnsemble / ShardedMembers on HIP tensors, stream ordering
between RCCL's kernels, MIOpen's and ours.  (The arithmetic of the sharded layouts at world sizes 2 and 3 is covered by
the gloo tests of tests/test_distributed.py; 8-GPU numbers come from the driver's bench.py --gpus 8 run.)
Runs in a child process so the process group does not outlive the test."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["TA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["TA_ROOT"], "tests"))
import transferattack_amd as ta
from transferattack_amd import _hip, backbones, dist as tadist
from transferattack_amd.utils import EnsembleModel, wrap_model
from conftest import u8_images

rank, world = tadist.init("nccl", single_rank_group=True)
assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
grp, member, shard, nshards = tadist.model_groups(world, 1)
x = u8_images(4, 224, 5).float() / 255
y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))

def make(name, build, model_name="injected", **kw):
    base = ta.load_attack_class(name)
    atk = type("R" + base.__name__, (base,), {"load_model": lambda self, mn: build()})(model_name=model_name, **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    return atk

def net():
    return wrap_model(backbones.create("toy_cnn", seed=3, verbose=False).eval().to(dev))

# device-tensor collectives through RCCL, as the sharded layouts issue them
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t, group=grp); dist.broadcast(t, src=0, group=grp)
parts = [torch.empty_like(t)]; dist.all_gather(parts, t, group=grp)
assert torch.equal(parts[0].cpu(), torch.arange(8, dtype=torch.float32))

for name, kw in (("ens", dict(epoch=3)), ("dim", dict(epoch=3))):
    tadist.seed_batch(5, 0)
    before = dict(_hip.stats)
    sharded = make(name, lambda: tadist.ShardedEnsemble(net(), grp, 1), **kw)(x, y)
    if name == "ens":      # the gradient went through the RCCL all-reduce: the local member's |g| sums must not be used
        assert _hip.stats["partials_reused"] == before["partials_reused"], "stale |g| sums used after the all-reduce"
    tadist.seed_batch(5, 0)
    plain = make(name, lambda: EnsembleModel([net()]), **kw)(x, y)
    assert sharded.is_cuda and torch.equal(sharded, plain), name
for name, kw in (("cwa", dict(epoch=2)), ("svre", dict(epoch=2)), ("adaea", dict(epoch=2))):
    tadist.seed_batch(5, 0)
    sharded = make(name, lambda: tadist.ShardedMembers(net(), member, grp, [0]), model_name=["a"], **kw)
    sharded.noise_source = (lambda s, lo, hi: torch.randn(s)) if name == "adaea" else sharded.noise_source
    d1 = sharded(x, y)
    tadist.seed_batch(5, 0)
    plain = make(name, lambda: EnsembleModel([net()]), model_name=["a"], **kw)
    plain.noise_source = sharded.noise_source
    d2 = plain(x, y)
    assert torch.equal(d1, d2), name
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("rccl world-of-one ok")
'''


def test_sharded_layouts_over_rccl_world_of_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl world-of-one ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bench_self_launch_refuses_missing_gpus():
    """bench.py --gpus N starts N ranks itself; with fewer devices than N it fails loudly instead of reporting n_gpus 1"""
    import torch
    need = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(need), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "only %d HIP device" % (need - 1) in (out.stderr + out.stdout)


def _world2(backend, extra_env=None, timeout=900):
    """two ranks of tests/tools/world2_child.py; -> their (returncode, output) pairs"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, TA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), TA_W2_BACKEND=backend,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "world2_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    out = []
    for p in procs:
        try:
            text, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            text, _ = p.communicate()
        out.append((p.returncode, text))
    return out


def test_world2_over_rccl():
    """The N > 1 paths on real devices (review: RCCL never saw more than one rank): two ranks, one GPU each, over RCCL -- image
    shards gathered equal the one-process loop, ShardedEnsemble with two distinct members equals the single-device
    EnsembleModel.  Needs two visible HIP devices: SKIPPED on the one-GPU box (tests/test_distributed.py runs the very same
    script over gloo on the kernels' host build), runs as is on the driver's 8-GPU node."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices, %d visible" % torch.cuda.device_count())
    for rank, (rc, text) in enumerate(_world2("nccl")):
        assert rc == 0 and "world-2 rank %d ok (nccl)" % rank in text, text[-4000:]
