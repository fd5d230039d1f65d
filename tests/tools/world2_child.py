"""TEST INFRASTRUCTURE -- one rank of the world-2 readiness check (tests/test_hip_rccl.py over RCCL on two GPUs;
tests/test_distributed.py runs the same script over gloo on the kernels' host build, so its logic is exercised where there
is one GPU or none).  Env: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*, TA_ROOT, TA_W2_BACKEND = nccl | gloo.

  (1) image shards (SURVEY 8e, configs[1]-[3]): 4 reference batches of a DIM attack (one geometry per batch, drawn on the
      host) split round-robin over the two ranks with per-batch seeding, no data-path collective -- gathered, they equal the
      one-process loop over all four batches, bit for bit;
  (2) ShardedEnsemble (configs[4]): two DISTINCT members, one per rank, logits and input gradients all-reduced -- the
      perturbation equals the single-device EnsembleModel's (a two-term sum has one rounding order)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.environ["TA_ROOT"]
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
backend = os.environ.get("TA_W2_BACKEND", "nccl")
if backend == "gloo":                                   # CPU twin: the kernels' host build in every rank
    import host_kernels

    class _Patch:
        def setattr(self, obj, name, value):
            setattr(obj, name, value)

        def setenv(self, name, value):
            os.environ[name] = value
    torch.set_num_threads(2)
    host_kernels.install(_Patch())
import transferattack_amd as ta                         # noqa: E402
from transferattack_amd import backbones, dist as tadist    # noqa: E402
from transferattack_amd.utils import EnsembleModel, wrap_model    # noqa: E402
from conftest import u8_images                          # noqa: E402

rank, world = tadist.init(backend)
assert world == 2 and dist.get_backend() == backend
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"])) if backend == "nccl" else torch.device("cpu")
size = 224 if backend == "nccl" else 64


def make(name, build, model_name="injected", **kw):
    base = ta.load_attack_class(name)
    atk = type("W" + base.__name__, (base,), {"load_model": lambda self, mn: build()})(model_name=model_name, **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    return atk


def net(seed):
    return wrap_model(backbones.create("toy_cnn", seed=seed, verbose=False).eval().to(dev))


def gather(t):
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    return parts


# (1) image shards: batches 0..3 of 3 images; rank r attacks batches r, r + 2
images = u8_images(12, size, 3).float() / 255
labels = torch.randint(0, 10, (12,), generator=torch.Generator().manual_seed(4))
atk = make("dim", lambda: net(3), epoch=3)
mine = {}
for b in tadist.shard_batches(4, rank, world):
    tadist.seed_batch(7, b)
    mine[b] = atk(images[3 * b:3 * b + 3], labels[3 * b:3 * b + 3])
stacked = gather(torch.stack([mine[b] for b in sorted(mine)]))          # [2, 3, 3, size, size] per rank
if rank == 0:
    for b in range(4):
        tadist.seed_batch(7, b)
        want = atk(images[3 * b:3 * b + 3], labels[3 * b:3 * b + 3])
        got = stacked[b % 2][b // 2].to(want.device)
        assert torch.equal(got, want), "image shard %d differs from the one-process run" % b
dist.barrier()

# (2) one DISTINCT surrogate per rank against the single-device ensemble of both
grp, member, shard, nshards = tadist.model_groups(world, 2)
assert (member, shard, nshards) == (rank, 0, 1)
x, y = images[:4], labels[:4]
tadist.seed_batch(5, 0)
sharded = make("ens", lambda: tadist.ShardedEnsemble(net(3 + rank), grp, 2), model_name=["a", "b"], epoch=3)(x, y)
both = gather(sharded)
assert torch.equal(both[0], both[1]), "the two ranks of a model group ended with different perturbations"
if rank == 0:
    tadist.seed_batch(5, 0)
    plain = make("ens", lambda: EnsembleModel([net(3), net(4)]), model_name=["a", "b"], epoch=3)(x, y)
    diff = float((plain != sharded).float().mean())
    assert diff == 0.0, "ShardedEnsemble differs from the single-device EnsembleModel in %.4f%% of the elements" % (100 * diff)
if dev.type == "cuda":
    torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("world-2 rank %d ok (%s)" % (rank, backend))
