"""CPU, build container only: the REAL reference, imported live from /root/reference through oracle/ref_shim.py, against
(a) the committed golden vectors -- so the fixtures under tests/golden/ are shown to be what the reference computes today,
not only what it computed when they were written -- and (b) the oracle restatement and the product's host-logic tier on
FRESH inputs (seeds no fixture uses), so the pin does not rest on the stored vectors alone.

Skipped where /root/reference is absent (the GPU box): nothing under `-m gpu`, smoke() or bench.py reads the reference."""
import os
import sys

import numpy as np
import pytest
import torch

import fake_hip
import fgsm_oracle as O
import transferattack_amd as ta
from conftest import u8_images
from transferattack_amd import backbones
from transferattack_amd.utils import wrap_model

REFERENCE = "/root/reference/transferattack"
pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is only present in the build container")


@pytest.fixture(scope="module")
def ref_shim():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ref_shim as shim
    shim.neutralise_cuda_calls()
    return shim


def t(a):
    return torch.from_numpy(np.asarray(a))


def toy():
    return backbones.create("toy_cnn", seed=3, verbose=False)


@pytest.mark.parametrize("name", ["mifgsm", "nifgsm", "vmifgsm", "dim", "tim", "sim", "admix"])
def test_goldens_are_what_the_reference_computes(golden, ref_shim, name):
    """re-run the reference's own class with the generator's seeds: the stored loop is reproduced bit for bit"""
    g = golden("loops_toy")
    x, label = t(g["x_u8"]).float() / 255, t(g["label"])
    atk = ref_shim.make_reference_attack(name, toy())
    torch.manual_seed(1234)
    assert np.array_equal(atk(x, label).numpy(), g["delta_" + name])


FRESH = [("mifgsm", {}), ("dim", {}), ("tim", {}), ("admix", {}), ("vmifgsm", dict(num_neighbor=3)), ("ifgssm", {}),
         ("rap", dict(epoch=5, transpoint=2, adv_steps=2)), ("usmm", dict(num_scale=2, num_mix=2))]
# ... and every other attack of the zoo that runs on a 32-pixel batch with one surrogate (40 s in all)
FRESH_FULL = FRESH + [("fgsm", {}), ("ifgsm", {}), ("nifgsm", {}), ("vnifgsm", dict(num_neighbor=3)), ("pifgsm", {}), ("emifgsm", {}),
                      ("iefgsm", {}), ("gra", dict(num_neighbor=3)), ("gnp", {}), ("pgn", dict(num_neighbor=3)), ("gifgsm", {}),
                      ("dta", dict(K=3)), ("pcifgsm", {}), ("smifgrm", dict(num_neighbor=3)), ("mig", dict(s_factor=4)),
                      ("aifgtm", {}), ("mef", dict(num_neighbor=3, epoch=4)), ("gaa", dict(N=3, epoch=4)), ("vaifgsm", dict(epoch=3)),
                      ("adamsi_fgm", {}), ("rgmifgsm", dict(num_directions=2, pre_epoch=2, epoch=3)), ("dual_mifgsm", dict(epoch=4)),
                      ("ens_mifgsm", dict(epoch=3, num_d=2)), ("foolmix", dict(epoch=3, m=2, n=2, k=3, print_timing=False)),
                      ("sim", {}), ("sia", dict(num_scale=3)), ("bsr", dict(num_scale=3)), ("dem", {}), ("maskblock", dict(patch_size=16)),
                      ("decowa", dict(num_warping=2, epoch=3)), ("ops", dict(num_sample_neighbor=2, num_sample_operator=2, epoch=2))]


@pytest.mark.parametrize("name,kw", FRESH_FULL)
def test_fresh_inputs_reference_vs_oracle_and_product(ref_shim, monkeypatch, name, kw):
    """inputs no fixture holds: reference (live) == oracle restatement (where it has a recipe) == product on the
    host-logic tier, bit for bit"""
    import random
    n, size = 3, 32
    x = u8_images(n, size, 555).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(556))

    def seed():
        random.seed(42); np.random.seed(42); torch.manual_seed(4242)

    ref = ref_shim.make_reference_attack(name, toy(), **kw)
    if name == "vaifgsm":
        ref.num_classes = 10
    seed()
    want = ref(x, label).detach().numpy()
    if name in O.RECIPES and name != "bsr":               # (the oracle's bsr recipe is pinned by its own golden test)
        seed()
        assert np.array_equal(O.run_attack(name, toy(), x, label, **kw).numpy(), want)
    fake_hip.install(monkeypatch)
    base = ta.load_attack_class(name)
    cls = type("Cpu" + base.__name__, (base,), {"load_model": lambda self, mn: wrap_model(toy().eval())})
    atk = cls(model_name="injected", **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    if name == "vaifgsm":
        atk.num_classes = 10
    seed()
    assert np.array_equal(atk(x, label).numpy(), want)


def test_l2t_operations_against_the_reference(ref_shim, monkeypatch):
    """every one of L2T's 98 candidate operations (l2t.py:375-387), forward and backward, against the reference's own
    operation object on the same input with the same seeds of all three generators: the operations that run as HIP kernels
    (sim, dim, admix, ssm -- here through the oracle-backed fake binding) and the device-op ones (rotations, block
    shuffle, drop-out, masks, crops, translations) give the reference's bytes"""
    import importlib
    import random
    ref_shim.import_reference()
    ref_ops = importlib.import_module("transferattack.input_transformation.l2t").op_list
    fake_hip.install(monkeypatch)
    from transferattack_amd.input_transformation import l2t as mine
    assert len(mine.op_list) == len(ref_ops) == 98
    gen = torch.Generator().manual_seed(77)
    x = torch.rand(2, 3, 224, 224, generator=gen)
    for k, (theirs, ours) in enumerate(zip(ref_ops, mine.op_list)):
        outs = []
        for op, host_draws in ((theirs, None), (ours, None), (ours, 'cpu')):
            monkeypatch.setattr(mine, "_draw_device", host_draws)          # None: product mode; 'cpu': seeded parity mode
            random.seed(k); np.random.seed(k); torch.manual_seed(k)
            xin = x.clone().requires_grad_(True)
            y = op(xin)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1000 + k))
            gx = torch.autograd.grad(y, xin, gy)[0] if y.requires_grad else torch.zeros_like(x)
            outs.append((y.detach(), gx))
        assert outs[0][0].shape == outs[1][0].shape, k
        for got in outs[1:]:
            assert torch.equal(outs[0][0], got[0]), "operation %d (%s): forward differs" % (k, type(ours).__name__)
            assert torch.equal(outs[0][1], got[1]), "operation %d (%s): backward differs" % (k, type(ours).__name__)
