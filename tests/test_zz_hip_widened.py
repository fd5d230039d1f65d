"""GPU (-m gpu), collected last: AdaEA / SMER / FGSRA on MI355X against the reference's golden loops.

These three attacks were added after the GPU minutes of their round were spent; until this file has run on a GPU box
their evidence is the CPU tiers (tests/test_host_logic.py bit for bit; tests/test_attack_loops_host.py, which runs
these very functions on the host stand-in).  The file sorts last so that a surprise here cannot mask the other tiers."""
import numpy as np
import pytest
import torch

import fgsm_oracle as O
import transferattack_amd as ta
from transferattack_amd import backbones
from transferattack_amd.utils import EnsembleModel, quantize_images, wrap_model

pytestmark = pytest.mark.gpu
EPS = 16 / 255
DEV = "cuda"
BOUND = 0.05            # uint8 mismatch vs the reference's golden images, as for the other end-to-end GPU tests


def t(a):
    return torch.from_numpy(np.asarray(a))


def mismatch(x, delta, ref_delta):
    return float((quantize_images(x, delta) != O.quantize_u8(x + t(ref_delta))).mean())


@pytest.mark.parametrize("name", ["adaea", "smer"])
def test_adaptive_ensembles(golden, name, batches=2):
    """the reference's loops on three members, consecutive batches (SMER's member weights persist on the object)"""
    g, base = golden("loops_ens"), golden("loops_toy")
    label = t(base["label"])
    inputs = [(t(base["x_u8"]).float() / 255, "delta_"), (t(g["x2_u8"]).float() / 255, "delta2_")][:batches]
    models = [backbones.create("toy_cnn", seed=s, verbose=False) for s in (3, 4, 5)]
    cls = ta.load_attack_class(name)

    def load_model(self, model_name):
        return EnsembleModel([wrap_model(m.eval().to(DEV)) for m in models])

    atk = type("Dev" + cls.__name__, (cls,), {"load_model": load_model})(model_name=["a", "b", "c"])
    if name == "adaea":
        atk.noise_source = lambda shape, lo, hi: torch.randn(shape)                    # reference's CPU draws
    else:
        atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    torch.manual_seed(1234)
    np.random.seed(99)
    for x, key in inputs:
        delta = atk(x, label).cpu()
        assert float(delta.abs().max()) <= EPS + 1e-7
        adv = x + delta
        assert float(adv.min()) >= 0.0 and float(adv.max()) <= 1.0 + 1e-7
        rate = mismatch(x, delta, g[key + name])
        print("%s %s: uint8 mismatch vs the reference's golden loop %.4f%%" % (name, key, 100 * rate))
        assert rate <= BOUND
    if name == "smer":
        learnt = atk.weight_selection.weight.detach().cpu().numpy()
        assert not np.array_equal(learnt, np.ones(3, dtype=np.float32))               # SGD inside the attack ran
        if batches == 2:
            np.testing.assert_allclose(learnt, g["smer_weight"], rtol=2e-2)       # 240 SGD steps; device rounding


def test_fgsra(golden):
    """DCT-domain neighbours (torch.fft on the device), relevance weighting, per-element step"""
    g, base = golden("loops_ens"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    cls = ta.load_attack_class("fgsra")
    model = backbones.create("toy_cnn", seed=3, verbose=False)
    atk = type("DevFGSRA", (cls,), {"load_model": lambda self, mn: wrap_model(model.eval().to(DEV))})(
        model_name="injected", max_iter=4)
    atk.noise_source = lambda shape, lo, hi: torch.rand(shape)
    probe = t(g["dct_probe"]).to(DEV)
    np.testing.assert_allclose(atk.dct_2d(probe).cpu().numpy(), g["dct_2d"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(atk.idct_2d(atk.dct_2d(probe)).cpu().numpy(), g["dct_probe"], atol=1e-5)
    torch.manual_seed(1234)
    delta = atk(x, label).cpu()
    assert float(delta.abs().max()) <= EPS + 1e-7
    rate = mismatch(x, delta, g["delta_fgsra"])
    print("fgsra: uint8 mismatch vs the reference's golden loop %.4f%%" % (100 * rate))
    assert rate <= BOUND
