"""CPU: the ASR fixtures written by oracle/gen_asr1000.py (the real reference's run) are what tests/test_hip_asr1000.py
assumes -- shapes, value ranges, the synthetic image set, and labels that ARE the surrogate's clean predictions (re-derived
here for the first images with the oracle's surrogate forward)."""
import os

import numpy as np
import pytest
import torch

import fgsm_oracle as O
from conftest import GOLDEN_DIR, u8_images
from transferattack_amd import backbones

CONFIGS = ("mifgsm", "dts", "ens", "vmifgsm")


@pytest.mark.parametrize("config", CONFIGS)
def test_asr_fixture_is_consistent(config):
    path = os.path.join(GOLDEN_DIR, "asr1000_%s.npz" % config)
    if not os.path.isfile(path):
        pytest.skip("fixture not generated yet (oracle/gen_asr1000.py %s)" % config)
    g = np.load(path)
    n, victims = int(g["n_images"]), [str(v) for v in g["victims"]]
    assert g["label"].shape == (n,) and g["clean_pred"].shape == (len(victims), n) and g["adv_pred"].shape == (len(victims), n)
    assert int(g["batch"]) == 32 and 0 <= g["label"].min() and g["label"].max() < 1000
    k = int(g["sign_images"])
    assert g["sign_bits"].size * 8 >= k * 3 * 224 * 224
    # labels = clean prediction of the surrogate(s) on the synthetic images the GPU test regenerates
    members = [backbones.create(s.split(":")[0], seed=int(s.split(":")[1]), verbose=False) for s in str(g["surrogate"]).split(",")]
    x = u8_images(4, 224, int(g["seed_images"])).float() / 255
    with torch.no_grad():
        pred = O.logits_of(members if len(members) > 1 else members[0], x).argmax(1).numpy()
    assert np.array_equal(pred, g["label"][:4].astype(np.int64))
    # the rates the fixture implies are informative for at least three victims (neither 0 nor 100 %)
    rates = (g["adv_pred"] != g["clean_pred"]).mean(axis=1)
    assert int(((rates > 0.05) & (rates < 0.95)).sum()) >= 3, rates


@pytest.mark.parametrize("config", ("mifgsm", "dts"))
def test_trained_victims_fixture_is_consistent(config):
    """tests/golden/asr_trained_<config>.npz + trained_toys.npz (oracle/gen_asr_trained.py): the committed weights classify the
    regenerated clean test images correctly (so main.py:90's literal rate is a transfer rate), the image set is the one the
    reference attacked (CRC), and the reference's rates are informative for the three held-out victims."""
    import gen_asr_trained as T
    path = os.path.join(GOLDEN_DIR, "asr_trained_%s.npz" % config)
    if not (os.path.isfile(path) and os.path.isfile(T.TOYS)):
        pytest.skip("fixture not generated yet (oracle/gen_asr_trained.py %s)" % config)
    g = np.load(path)
    n, nets = int(g["n_images"]), [str(v) for v in g["nets"]]
    assert nets == list(T.NETS) and g["adv_pred"].shape == (len(nets), n) and int(g["batch"]) == 32
    xu8, label = T.make_images(64, int(g["seed_images"]))          # (the GPU test regenerates all 1000 and checks their CRC)
    assert np.array_equal(label.numpy(), g["label"][:64].astype(np.int64))
    for v, name in enumerate(nets):
        pred = T.predict(T.load_trained(name), xu8.float() / 255).numpy()
        assert np.array_equal(pred, g["clean_pred"][v][:64].astype(np.int64)) and (pred == label.numpy()).mean() >= 0.99
    rates = (g["adv_pred"].astype(np.int64) != g["label"].astype(np.int64)[None]).mean(axis=1)
    assert int(((rates[1:] > 0.05) & (rates[1:] < 0.97)).sum()) >= 2, rates
