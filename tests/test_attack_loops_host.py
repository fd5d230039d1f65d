"""Whole attack loops on the CPU THROUGH THE KERNEL SOURCES: the product's attack classes, the real ctypes binding and
the .hip kernels compiled for the host (tests/hipcpu, tests/host_kernels.py); the surrogate runs on torch's CPU path,
i.e. with the reference's own arithmetic.  The assertions are those of the `-m gpu` loop tests (tests/test_hip_attacks.py)
re-used function by function.

Compared with tests/test_host_logic.py (binding replaced by an oracle-backed fake) this tier executes the kernels'
code; compared with the GPU tier it lacks the gfx950 build and MIOpen's rounding.  With the surrogate's arithmetic
equal to the reference's, the only admissible difference from the golden loops is the fixed-order sum|g| of the update
kernels, so the uint8 mismatch bound of the GPU tests is met with a wide margin.  Test infrastructure."""
import numpy as np
import pytest
import torch

import host_kernels
import test_hip_attacks as A
from transferattack_amd import _hip


@pytest.fixture(autouse=True)
def host_backend(monkeypatch):
    host_kernels.install(monkeypatch)
    monkeypatch.setattr(A, "DEV", "cpu")


import os

FULL = os.environ.get("TA_HOST_FULL", "0") == "1"        # every case of the GPU tier (about 3.5 min on 8 cores)


def _subset(cases, keep):
    return cases if FULL else [c for c in cases if (c[0] if isinstance(c, tuple) else c) in keep]


test_config1_trajectory_replay = A.test_config1_trajectory_replay
test_variants_run = A.test_variants_run_on_gpu


@pytest.mark.parametrize("name", _subset(["mifgsm", "nifgsm", "tim", "sim", "admix", "dim", "dts"],
                                         {"mifgsm", "nifgsm", "tim", "sim", "dim", "dts"}))
def test_trajectory_replay_reference_gradients(golden, name):
    A.test_trajectory_replay_reference_gradients(golden, name)


@pytest.mark.parametrize("name", _subset(["fgsm", "ifgsm", "mifgsm", "nifgsm", "vmifgsm", "vnifgsm", "dim", "tim", "sim",
                                          "admix", "dts", "ens"], {"fgsm", "mifgsm", "dim", "tim", "sim", "dts", "ens"}))
def test_end_to_end_vs_reference(golden, name):
    A.test_end_to_end_gpu_vs_reference(golden, name)


@pytest.mark.parametrize("name,kw", _subset([("pifgsm", {}), ("emifgsm", {}), ("iefgsm", {}), ("gnp", {}),
                                             ("gra", dict(num_neighbor=5)), ("pgn", dict(num_neighbor=4)), ("gifgsm", {}),
                                             ("dta", dict(K=3)), ("pcifgsm", {}), ("smifgrm", dict(num_neighbor=4))],
                                            {"pifgsm", "iefgsm", "gra", "gifgsm", "dta", "pcifgsm"}))
def test_more_gradient_attacks_vs_reference(golden, name, kw):
    A.test_more_gradient_attacks_gpu_vs_reference(golden, name, kw)


@pytest.mark.parametrize("name", _subset(["svre", "cwa"], {"cwa"}))
def test_per_member_ensemble_attacks(golden, name):
    A.test_per_member_ensemble_attacks_gpu(golden, name)


def _mismatch(x, delta, ref_delta):
    from transferattack_amd.utils import quantize_images
    return float((quantize_images(x, delta) != A.O.quantize_u8(x + A.t(ref_delta))).mean())


@pytest.mark.parametrize("name", ["adaea", "smer"])
def test_adaptive_ensembles_through_kernels(golden, name):
    """AdaEA / SMER (added after the GPU minutes of their round were spent): the reference's golden loops on three
    members, two batches in a row, with the update kernels' own code in the loop."""
    from transferattack_amd import backbones
    from transferattack_amd.utils import EnsembleModel, wrap_model
    import transferattack_amd as ta
    g, base = golden("loops_ens"), golden("loops_toy")
    x, label = A.t(base["x_u8"]).float() / 255, A.t(base["label"])
    x2 = A.t(g["x2_u8"]).float() / 255
    models = [backbones.create("toy_cnn", seed=s, verbose=False) for s in (3, 4, 5)]
    cls = ta.load_attack_class(name)
    atk = type("Host" + cls.__name__, (cls,), {
        "load_model": lambda self, mn: EnsembleModel([wrap_model(m.eval()) for m in models])})(model_name=["a", "b", "c"])
    atk.noise_source = (lambda shape, lo, hi: torch.randn(shape)) if name == "adaea" else (
        lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi))
    torch.manual_seed(1234)
    np.random.seed(99)
    for batch, key in ((x, "delta_"), (x2, "delta2_"))[:2 if FULL or name == "adaea" else 1]:
        delta = atk(batch, label)
        assert float(delta.abs().max()) <= A.EPS + 1e-7
        rate = _mismatch(batch, delta, g[key + name])
        print("%s %s: uint8 mismatch vs the reference's golden loop %.4f%%" % (name, key, 100 * rate))
        assert rate <= 0.05


def test_fgsra_through_kernels(golden):
    g, base = golden("loops_ens"), golden("loops_toy")
    x, label = A.t(base["x_u8"]).float() / 255, A.t(base["label"])
    atk = A.make("fgsra", max_iter=4)
    atk.noise_source = lambda shape, lo, hi: torch.rand(shape)
    torch.manual_seed(1234)
    delta = atk(x, label)
    rate = _mismatch(x, delta, g["delta_fgsra"])
    print("fgsra: uint8 mismatch vs the reference's golden loop %.4f%%" % (100 * rate))
    assert float(delta.abs().max()) <= A.EPS + 1e-7 and rate <= 0.05
