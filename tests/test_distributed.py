"""CPU, world_size = 2, gloo: the N>1 paths -- image sharding by whole batches with per-batch seeding, and the
ensemble exchange (logit all-reduce + input-gradient all-reduce) -- against the single-process result.
The kernel binding is replaced by the oracle-backed fake in every rank (tests/fake_hip.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def same_thread_count_as_the_ranks():
    """the single-process side of every comparison runs with the ranks' thread count: ATen's CPU bilinear resize (the
    32 -> 224 resize in front of the toy surrogate) can round differently for different work partitions"""
    threads = torch.get_num_threads()
    torch.set_num_threads(2)
    yield
    torch.set_num_threads(threads)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Patch:
    """minimal stand-in for pytest's monkeypatch inside a spawned rank"""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)

    def setenv(self, name, value):
        os.environ[name] = value


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import fake_hip
    fake_hip.install(_Patch())
    from transferattack_amd import dist as tadist
    tadist.init("gloo")
    return tadist


def _make(name, models, model_name="injected", **kw):
    import transferattack_amd as ta
    from transferattack_amd.utils import EnsembleModel, wrap_model
    base = ta.load_attack_class(name)

    def load_model(self, model_name):
        if isinstance(models, torch.nn.Module) and not isinstance(models, torch.nn.Sequential):
            return models                       # a ready ShardedEnsemble
        wrapped = [wrap_model(m.eval()) for m in models]
        return wrapped[0] if len(wrapped) == 1 else EnsembleModel(wrapped)

    return type("T" + base.__name__, (base,), {"load_model": load_model})(model_name=model_name, **kw)


def _rank_shard(rank, world, port, out):
    tadist = _setup(rank, world, port)
    from transferattack_amd import backbones
    from conftest import u8_images
    images = u8_images(20, 32, 3).float() / 255             # 20 images in batches of 8 -> 8, 8, 4
    labels = torch.randint(0, 10, (20,), generator=torch.Generator().manual_seed(4))
    atk = _make("dim", [backbones.create("toy_cnn", seed=3, verbose=False)], epoch=3)
    result = {}
    for b in tadist.shard_batches(3, rank, world):
        tadist.seed_batch(99, b)
        sl = slice(8 * b, min(8 * b + 8, 20))
        result[b] = atk(images[sl], labels[sl]).numpy()
    gathered = [None] * world
    dist.all_gather_object(gathered, result)
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        np.savez(out, **{"b%d" % k: v for k, v in merged.items()})
    dist.barrier()
    dist.destroy_process_group()


def _rank_ensemble(rank, world, port, out):
    tadist = _setup(rank, world, port)
    from transferattack_amd import backbones
    from transferattack_amd.utils import wrap_model
    from conftest import u8_images
    x = u8_images(4, 32, 5).float() / 255
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    grp, idx, shard, nshards = tadist.model_groups(world, 2)
    assert (idx, shard, nshards) == (rank, 0, 1)
    member = wrap_model(backbones.create("toy_cnn", seed=3 + rank, verbose=False).eval())
    atk = _make("ens", tadist.ShardedEnsemble(member, grp, 2), epoch=4)
    delta = atk(x, y)
    gathered = [None] * world
    dist.all_gather_object(gathered, delta.numpy())
    if rank == 0:
        np.savez(out, r0=gathered[0], r1=gathered[1])
    dist.barrier()
    dist.destroy_process_group()


def _rank_ensemble_ind(rank, world, port, out):
    """mode 'ind' (utils.py:101-103): the [M, N, classes] stack and the gradient of a loss that treats the members differently"""
    tadist = _setup(rank, world, port)
    from transferattack_amd import backbones
    from transferattack_amd.utils import wrap_model
    from conftest import u8_images
    x = (u8_images(4, 32, 5).float() / 255).requires_grad_(True)
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    grp, idx, _, _ = tadist.model_groups(world, 2)
    member = wrap_model(backbones.create("toy_cnn", seed=3 + rank, verbose=False).eval())
    ens = tadist.ShardedEnsemble(member, grp, 2, mode='ind')
    stack = ens(x)
    loss = torch.nn.functional.cross_entropy(stack[0], y) + 3.0 * torch.nn.functional.cross_entropy(stack[1], y)
    grad = torch.autograd.grad(loss, x)[0]
    gathered = [None] * world
    dist.all_gather_object(gathered, (stack.detach().numpy(), grad.numpy()))
    if rank == 0:
        np.savez(out, s0=gathered[0][0], g0=gathered[0][1], s1=gathered[1][0], g1=gathered[1][1])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ensemble_mode_ind(tmp_path, monkeypatch):
    """ShardedEnsemble(mode='ind') on 2 ranks == EnsembleModel(mode='ind') in one process: same stack, same input gradient"""
    got = _run(_rank_ensemble_ind, tmp_path)
    assert np.array_equal(got["s0"], got["s1"]) and np.array_equal(got["g0"], got["g1"])
    import fake_hip
    fake_hip.install(monkeypatch)
    from transferattack_amd import backbones
    from transferattack_amd.utils import EnsembleModel, wrap_model
    from conftest import u8_images
    x = (u8_images(4, 32, 5).float() / 255).requires_grad_(True)
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    ens = EnsembleModel([wrap_model(backbones.create("toy_cnn", seed=3 + k, verbose=False).eval()) for k in range(2)], mode='ind')
    stack = ens(x)
    loss = torch.nn.functional.cross_entropy(stack[0], y) + 3.0 * torch.nn.functional.cross_entropy(stack[1], y)
    grad = torch.autograd.grad(loss, x)[0]
    assert stack.shape == (2, 4, 10)
    np.testing.assert_allclose(got["s0"], stack.detach().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["g0"], grad.numpy(), rtol=1e-5, atol=1e-7)


def _rank_ensemble_bsr(rank, world, port, out):
    """a transform attack that draws from Python's ``random`` (BSR: axis order, strips, permutations) on a model list:
    every rank of the group must apply the SAME transform or the all-reduced gradient mixes pixel arrangements"""
    tadist = _setup(rank, world, port)
    from transferattack_amd import backbones
    from transferattack_amd.utils import wrap_model
    from conftest import u8_images
    x = u8_images(2, 32, 5).float() / 255
    y = torch.randint(0, 10, (2,), generator=torch.Generator().manual_seed(6))
    grp, idx, _, _ = tadist.model_groups(world, 2)
    import random
    random.seed(1000 + rank)                                   # whatever state the ranks arrive with, it differs
    member = wrap_model(backbones.create("toy_cnn", seed=3 + rank, verbose=False).eval())
    atk = _make("bsr", tadist.ShardedEnsemble(member, grp, 2), model_name=["a", "b"], epoch=2, num_scale=3)
    tadist.seed_batch(5, 0)
    delta = atk(x, y)
    gathered = [None] * world
    dist.all_gather_object(gathered, delta.numpy())
    if rank == 0:
        np.savez(out, r0=gathered[0], r1=gathered[1])
    dist.barrier()
    dist.destroy_process_group()


def _rank_members(rank, world, port, out):
    tadist = _setup(rank, world, port)
    from transferattack_amd import backbones
    from transferattack_amd.utils import wrap_model
    from conftest import u8_images
    x = u8_images(4, 32, 5).float() / 255
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    grp, idx, _, _ = tadist.model_groups(world, 2)
    result = {}
    for name, kw in _MEMBER_ATTACKS:
        member = wrap_model(backbones.create("toy_cnn", seed=3 + rank, verbose=False).eval())
        atk = _make(name, tadist.ShardedMembers(member, idx, grp, [0, 1]), model_name=["a", "b"], **kw)
        atk.noise_source = _cpu_noise(name)
        tadist.seed_batch(5, 0)
        calls = [0]
        plain = tadist.MemberHandle.forward

        def counted(self, v, _calls=calls, _plain=plain):
            _calls[0] += 1
            return _plain(self, v)

        tadist.MemberHandle.forward = counted
        result[name] = atk(x, y).numpy()
        tadist.MemberHandle.forward = plain
        # AdaEA evaluates all members side by side (all-gather / all-reduce rounds), the others address single members
        assert (calls[0] == 0) == (name == "adaea"), (name, calls[0])
    gathered = [None] * world
    dist.all_gather_object(gathered, result)
    if rank == 0:
        np.savez(out, **{"%s_r%d" % (k, r): v for r, g in enumerate(gathered) for k, v in g.items()})
    dist.barrier()
    dist.destroy_process_group()


_MEMBER_ATTACKS = (("svre", dict(epoch=2)), ("cwa", dict(epoch=2)), ("adaea", dict(epoch=2)), ("smer", dict(epoch=2)))


def _cpu_noise(name):
    if name == "adaea":
        return lambda shape, lo, hi: torch.randn(shape)
    return lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)


_CLI_CASES = (("dim", "toy_cnn"), ("ens", "toy_cnn,toy_cnn"), ("cwa", "toy_cnn,toy_cnn"))


def _write_dataset(root, count):
    import csv
    from PIL import Image
    from conftest import u8_images
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    pixels = u8_images(count, 32, 11).permute(0, 2, 3, 1).numpy()
    with open(os.path.join(root, "labels.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["filename", "label", "targeted_label"])
        for i in range(count):
            Image.fromarray(pixels[i]).save(os.path.join(root, "images", "%d.png" % i))
            w.writerow(["%d.png" % i, i % 10, (i + 1) % 10])


def _cli(inp, out, attack, model, *extra):
    import main as cli
    sys.argv = ["main.py", "--input_dir", inp, "--output_dir", out, "--attack", attack, "--model", model,
                "--batchsize", "2", "--seed", "5"] + list(extra)
    cli.main()


def _rank_cli(rank, world, port, out):
    """main.py under torch.distributed (gloo), the kernels' own code on the host (tests/host_kernels.py) in every rank"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import host_kernels
    host_kernels.install(_Patch())
    base = os.path.dirname(out)
    for attack, model in _CLI_CASES:
        _cli(os.path.join(base, "data"), os.path.join(base, "adv2_" + attack), attack, model)
        dist.barrier()
    dist.destroy_process_group()


def _rank_adaea3(rank, world, port, out):
    tadist = _setup(rank, world, port)
    from transferattack_amd import backbones
    from transferattack_amd.utils import wrap_model
    from conftest import u8_images
    x = u8_images(4, 32, 5).float() / 255
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    grp, idx, _, _ = tadist.model_groups(world, 3)
    member = wrap_model(backbones.create("toy_cnn", seed=3 + rank, verbose=False).eval())
    atk = _make("adaea", tadist.ShardedMembers(member, idx, grp, [0, 1, 2]), model_name=["a", "b", "c"], epoch=3)
    atk.noise_source = _cpu_noise("adaea")
    tadist.seed_batch(5, 0)
    delta = atk(x, y).numpy()
    gathered = [None] * world
    dist.all_gather_object(gathered, delta)
    if rank == 0:
        np.savez(out, r0=gathered[0], r1=gathered[1], r2=gathered[2])
    dist.barrier()
    dist.destroy_process_group()


def _run(fn, tmp_path, world=2):
    out = str(tmp_path / "out.npz")
    mp.spawn(fn, args=(world, _free_port(), out), nprocs=world, join=True)
    return np.load(out)


def test_image_sharding_is_gpu_count_invariant(tmp_path, monkeypatch):
    """2 ranks x round-robin batches == 1 process over all batches, bit for bit (per-batch seeding)."""
    got = _run(_rank_shard, tmp_path)
    import fake_hip
    fake_hip.install(monkeypatch)
    from transferattack_amd import backbones, dist as tadist
    from conftest import u8_images
    assert tadist.shard_batches(32, 3, 8) == [3, 11, 19, 27] and tadist.shard_batches(3, 1, 2) == [1]
    assert sorted(sum((tadist.shard_batches(32, r, 8) for r in range(8)), [])) == list(range(32))
    images = u8_images(20, 32, 3).float() / 255
    labels = torch.randint(0, 10, (20,), generator=torch.Generator().manual_seed(4))
    atk = _make("dim", [backbones.create("toy_cnn", seed=3, verbose=False)], epoch=3)
    for b in range(3):
        tadist.seed_batch(99, b)
        sl = slice(8 * b, min(8 * b + 8, 20))
        assert np.array_equal(atk(images[sl], labels[sl]).numpy(), got["b%d" % b]), b


def test_sharded_ensemble_matches_single_process(tmp_path, monkeypatch):
    """one member per rank + 2 all-reduces == EnsembleModel on one device (fp32 sums in a different order:
    gradients agree to rounding, the final perturbation to the few pixels rounding can flip)."""
    got = _run(_rank_ensemble, tmp_path)
    assert np.array_equal(got["r0"], got["r1"])                      # both ranks hold the same delta
    import fake_hip
    fake_hip.install(monkeypatch)
    from transferattack_amd import backbones
    from conftest import u8_images
    x = u8_images(4, 32, 5).float() / 255
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
    ref = _make("ens", models, epoch=4)(x, y).numpy()
    assert float((got["r0"] != ref).mean()) <= 0.002
    assert np.abs(got["r0"] - ref).max() <= 2 * 1.6 / 255 + 1e-7


def test_sharded_ensemble_bsr_same_draws_on_every_rank(tmp_path, monkeypatch):
    """BSR on a two-member list, one member per rank: ``seed_batch`` seeds Python's ``random`` too, so both ranks shuffle
    the same blocks -> r0 == r1, and equal to the single-process EnsembleModel run up to the rounding of the member sum."""
    got = _run(_rank_ensemble_bsr, tmp_path)
    assert np.array_equal(got["r0"], got["r1"])
    import fake_hip
    fake_hip.install(monkeypatch)
    from transferattack_amd import backbones, dist as tadist
    from conftest import u8_images
    x = u8_images(2, 32, 5).float() / 255
    y = torch.randint(0, 10, (2,), generator=torch.Generator().manual_seed(6))
    models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
    atk = _make("bsr", models, model_name=["a", "b"], epoch=2, num_scale=3)
    tadist.seed_batch(5, 0)
    ref = atk(x, y).numpy()
    assert float((got["r0"] != ref).mean()) <= 0.005
    # and without the seeding of ``random`` the ranks would have diverged: the draw really comes from that generator
    import random
    from transferattack_amd.transforms import bsr_draw
    tadist.seed_batch(5, 0)
    a = bsr_draw((2, 3, 32, 32), 3, 3)
    tadist.seed_batch(5, 0)
    same = bsr_draw((2, 3, 32, 32), 3, 3)
    tadist.seed_batch(5, 0)
    random.seed(78)                                            # torch and numpy as before, ``random`` elsewhere
    b = bsr_draw((2, 3, 32, 32), 3, 3)
    assert np.array_equal(a, same) and not np.array_equal(a, b)


def test_sharded_members_match_single_process(tmp_path, monkeypatch):
    """SVRE / CWA / AdaEA / SMER with one member per rank (logits broadcast forward, input gradient broadcast backward)
    == the same attack with both members in one process, bit for bit: the ranks run the same arithmetic, only the
    surrogates are spread out."""
    got = _run(_rank_members, tmp_path)
    import fake_hip
    fake_hip.install(monkeypatch)
    from transferattack_amd import backbones, dist as tadist
    from conftest import u8_images
    x = u8_images(4, 32, 5).float() / 255
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    for name, kw in _MEMBER_ATTACKS:
        models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
        atk = _make(name, models, model_name=["a", "b"], **kw)
        atk.noise_source = _cpu_noise(name)
        tadist.seed_batch(5, 0)
        ref = atk(x, y).numpy()
        assert np.array_equal(got[name + "_r0"], got[name + "_r1"]), name
        assert np.array_equal(got[name + "_r0"], ref), name


def test_main_cli_sharded_equals_single_process(tmp_path, monkeypatch):
    """main.py on 2 ranks writes the same PNG bytes as on 1: image sharding by whole batches with per-batch seeding
    (dim), one surrogate per rank with the two all-reduces (ens), per-member broadcast (cwa).  Runs the kernel
    sources on the host stand-in in every rank, gloo instead of RCCL."""
    from PIL import Image
    _write_dataset(str(tmp_path / "data"), 5)                      # 5 images, batches of 2 -> 3 batches over 2 ranks
    # (ens / cwa: the 2 ranks form ONE model group, so there the batches are not sharded but the surrogates are)
    mp.spawn(_rank_cli, args=(2, _free_port(), str(tmp_path / "out.npz")), nprocs=2, join=True)
    import host_kernels
    host_kernels.install(monkeypatch)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    for attack, model in _CLI_CASES:
        _cli(str(tmp_path / "data"), str(tmp_path / ("adv1_" + attack)), attack, model)
        for i in range(5):
            one = np.array(Image.open(tmp_path / ("adv1_" + attack) / ("%d.png" % i)))
            two = np.array(Image.open(tmp_path / ("adv2_" + attack) / ("%d.png" % i)))
            assert np.array_equal(one, two), (attack, i)
        assert not np.array_equal(one, np.array(Image.open(tmp_path / "data" / "images" / "4.png")))   # it did attack


def test_adaea_three_members_side_by_side(tmp_path, monkeypatch):
    """AdaEA with one member on each of 3 ranks (all-gather of logits / per-member gradients, all-reduce of the fused
    input gradient): every rank ends with the same delta; against the single-process run the fused gradient is a sum
    of three terms in a different order, so it agrees to rounding and the perturbation to the few pixels that can flip."""
    got = _run(_rank_adaea3, tmp_path, world=3)
    assert np.array_equal(got["r0"], got["r1"]) and np.array_equal(got["r0"], got["r2"])
    import fake_hip
    fake_hip.install(monkeypatch)
    from transferattack_amd import backbones, dist as tadist
    from conftest import u8_images
    x = u8_images(4, 32, 5).float() / 255
    y = torch.randint(0, 10, (4,), generator=torch.Generator().manual_seed(6))
    models = [backbones.create("toy_cnn", seed=3 + k, verbose=False) for k in range(3)]
    atk = _make("adaea", models, model_name=["a", "b", "c"], epoch=3)
    atk.noise_source = _cpu_noise("adaea")
    tadist.seed_batch(5, 0)
    ref = atk(x, y).numpy()
    assert float((got["r0"] != ref).mean()) <= 0.002
    assert np.abs(got["r0"] - ref).max() <= 2 * 1.6 / 255 + 1e-7


# ---- one DISTINCT surrogate per rank, 224-pixel inputs, the kernels' own code in every rank: at 224 px the
# backward of the preprocessing Normalize is the last kernel of the LOCAL member's backward and leaves |g| tile sums of the
# local gradient only; the summed gradient must not be normalised with them.  Also: every attack class -- not only ENS --
# gets the members' gradient sum when it runs on a ShardedEnsemble.
_DISTINCT_CASES = (("ens", dict(epoch=3)), ("dim", dict(epoch=3)), ("mifgsm", dict(epoch=2)))


def _rank_distinct_members(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import host_kernels
    host_kernels.install(_Patch())
    from transferattack_amd import _hip, backbones, dist as tadist
    from transferattack_amd.utils import wrap_model
    from conftest import u8_images
    tadist.init("gloo")
    x = u8_images(2, 224, 5).float() / 255
    y = torch.randint(0, 10, (2,), generator=torch.Generator().manual_seed(6))
    grp, idx, _, _ = tadist.model_groups(world, 2)
    result = {}
    for name, kw in _DISTINCT_CASES:
        member = wrap_model(backbones.create("toy_cnn", seed=3 + rank, verbose=False).eval())
        atk = _make(name, tadist.ShardedEnsemble(member, grp, 2), **kw)
        tadist.seed_batch(5, 0)
        before = _hip.stats["partials_reused"]
        result[name] = atk(x, y).numpy()
        # (DIM's backward runs AFTER the all-reduce, on the summed gradient: its sums are the right ones and are used)
        assert name == "dim" or _hip.stats["partials_reused"] == before, \
            "%s: stale |g| sums of the local member were used after the all-reduce" % name
    gathered = [None] * world
    dist.all_gather_object(gathered, result)
    if rank == 0:
        np.savez(out, **{"%s_r%d" % (k, r): v for r, g in enumerate(gathered) for k, v in g.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ensemble_distinct_members_at_224(tmp_path, monkeypatch):
    """two different members, one per rank, 224-px images, kernel sources on the host in every rank: both ranks end with
    the same perturbation, and it equals the single-process EnsembleModel run bit for bit (two-term sums commute) -- for
    ENS and for attacks that know nothing about the sharding (DIM, MI-FGSM on a model list)."""
    got = _run(_rank_distinct_members, tmp_path)
    import host_kernels
    host_kernels.install(monkeypatch)
    from transferattack_amd import backbones, dist as tadist
    from conftest import u8_images
    x = u8_images(2, 224, 5).float() / 255
    y = torch.randint(0, 10, (2,), generator=torch.Generator().manual_seed(6))
    for name, kw in _DISTINCT_CASES:
        models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
        tadist.seed_batch(5, 0)
        ref = _make(name, models, **kw)(x, y).numpy()
        assert np.array_equal(got[name + "_r0"], got[name + "_r1"]), name
        assert np.array_equal(got[name + "_r0"], ref), name


def test_main_cli_coalesced_batches_equal_reference_batches(tmp_path, monkeypatch, capsys):
    """main.py --coalesce: a batch-independent attack (MI-FGSM) run three reference batches per device batch writes the
    same PNG bytes as one reference batch at a time -- the per-image normalisation removes the 1/N of the batch-mean loss
    (a power of two here, so even the rounding is the same); a batch-coupled attack (DIM) is never coalesced."""
    from PIL import Image
    import json
    _write_dataset(str(tmp_path / "data"), 6)
    import host_kernels
    host_kernels.install(monkeypatch)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    lines = {}
    for tag, extra in (("k1", ["--coalesce", "1", "--profile"]), ("k2", ["--coalesce", "2", "--profile"]), ("auto", ["--profile"])):
        _cli(str(tmp_path / "data"), str(tmp_path / ("mi_" + tag)), "mifgsm", "toy_cnn", *extra)
        lines[tag] = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert [lines[t]["reference_batches_per_device_batch"] for t in ("k1", "k2", "auto")] == [1, 2, 4]
    assert all(lines[t]["images"] == 6 and lines[t]["end_to_end_images_per_s"] > 0 for t in lines)
    for i in range(6):
        one = np.array(Image.open(tmp_path / "mi_k1" / ("%d.png" % i)))
        assert np.array_equal(one, np.array(Image.open(tmp_path / "mi_k2" / ("%d.png" % i)))), i
    _cli(str(tmp_path / "data"), str(tmp_path / "dim_k4"), "dim", "toy_cnn", "--coalesce", "4", "--profile")
    line = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert line["reference_batches_per_device_batch"] == 1


def test_world2_readiness_script_over_gloo():
    """tests/tools/world2_child.py -- what tests/test_hip_rccl.py::test_world2_over_rccl runs on two GPUs over RCCL -- over
    gloo, the kernels from their host build in both ranks: the script's logic is exercised here, so the first node with
    two devices meets known-good code."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_hip_rccl as R
    for rank, (rc, text) in enumerate(R._world2("gloo", timeout=1200)):
        assert rc == 0 and "world-2 rank %d ok (gloo)" % rank in text, text[-4000:]
