"""The config-size parity tests of the GPU tier (tests/test_hip_configs.py), run on the CPU THROUGH THE KERNEL SOURCES
(tests/hipcpu, tests/host_kernels.py): the product's attack classes, the real binding and the .hip kernels compiled for
the host; the surrogate on torch's CPU path.  Same assertions, function by function.  A subset by default (each case
runs the surrogate twice on the CPU); TA_HOST_FULL=1 runs them all.  Test infrastructure."""
import os

import pytest

import host_kernels
import test_hip_configs as G

FULL = os.environ.get("TA_HOST_FULL", "0") == "1"


@pytest.fixture(autouse=True)
def host_backend(monkeypatch):
    host_kernels.install(monkeypatch)
    monkeypatch.setattr(G, "DEV", "cpu")


test_config2_mifgsm_resnet50_replay = G.test_config2_mifgsm_resnet50_replay


@pytest.mark.parametrize("tag,model,kw", [("vit", "vit_base_patch16_224", dict(num_neighbor=4, epoch=3))] +
                         ([("resnet18", "resnet18", dict(num_neighbor=20, epoch=3))] if FULL else []))
def test_config4_vmifgsm_replay(golden, monkeypatch, tag, model, kw):
    G.test_config4_vmifgsm_replay(golden, monkeypatch, tag, model, kw)


@pytest.mark.skipif(not FULL, reason="TA_HOST_FULL=1: DTS / four-member ensemble at 224 px run the surrogates twice on the CPU")
def test_config3_and_config5_replay(golden, monkeypatch):
    G.test_config3_dts_resnet50_replay(golden, monkeypatch)
    G.test_config5_ensemble_replay(golden, monkeypatch)
