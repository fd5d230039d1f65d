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

FULL = os.environ.get("TA_HOST_FULL", "1") == "1"        # every case of the GPU tier (about 1.7 min on 8 cores); 0 = a subset


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


@pytest.mark.parametrize("name", ["adaea", "smer"])
def test_adaptive_ensembles_through_kernels(golden, monkeypatch, name):
    """AdaEA / SMER: the functions of the (not yet run) GPU file tests/test_zz_hip_widened.py, on the host stand-in"""
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)                  # surrogate arithmetic = the reference's: nothing may differ
    W.test_adaptive_ensembles(golden, name, batches=2 if FULL or name == "adaea" else 1)


def test_fgsra_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)
    W.test_fgsra(golden)


def test_sia_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)
    W.test_sia_attack(golden)


def test_ssm_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)
    W.test_ssm_attack(golden)


def test_main_cli_roundtrip(tmp_path, monkeypatch):
    """main.py end to end (decode -> attack -> quantise -> PNG -> --eval) with the kernels' own code on the host"""
    A.test_main_cli_roundtrip(tmp_path, monkeypatch)


def test_main_cli_resume(tmp_path, monkeypatch):
    """--resume: an interrupted run (outputs of one batch missing) recomputes exactly the missing batch, and what it
    writes equals the uninterrupted run (per-batch seeding)"""
    import sys
    from PIL import Image
    import main as cli
    import test_distributed as T
    T._write_dataset(str(tmp_path / "data"), 5)
    argv = ["main.py", "--input_dir", str(tmp_path / "data"), "--output_dir", str(tmp_path / "adv"), "--attack", "dim",
            "--model", "toy_cnn", "--batchsize", "2", "--seed", "5"]
    monkeypatch.setattr(sys, "argv", argv)
    cli.main()
    full = {i: np.array(Image.open(tmp_path / "adv" / ("%d.png" % i))) for i in range(5)}
    stamp = {i: (tmp_path / "adv" / ("%d.png" % i)).stat().st_mtime_ns for i in range(5)}
    (tmp_path / "adv" / "2.png").unlink()                                   # batch 1 = images 2, 3 becomes incomplete
    monkeypatch.setattr(sys, "argv", argv + ["--resume"])
    cli.main()
    for i in range(5):
        assert np.array_equal(np.array(Image.open(tmp_path / "adv" / ("%d.png" % i))), full[i])
    after = {i: (tmp_path / "adv" / ("%d.png" % i)).stat().st_mtime_ns for i in range(5)}
    assert [i for i in range(5) if after[i] != stamp[i]] == [2, 3]          # only the incomplete batch was redone


def test_tim_loop_with_separable_smoothing(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)
    W.test_tim_loop_with_separable_smoothing(golden, monkeypatch)
