"""GPU (-m gpu): the end-to-end parity statement of BASELINE.json -- attack success rate on the 1000-image set.

``oracle/gen_asr1000.py`` ran the REAL reference (its own MI-FGSM class / its own DIM, TIM and SIM methods composed as
SURVEY.md a17) on the CPU over 1000 seeded synthetic images in the reference's 32-image batches and stored, per image
and per victim, the prediction on the clean and on the adversarial image (tests/golden/asr1000_<config>.npz).  Here the
PRODUCT runs the same job on MI355X -- same images, labels, seeded weights, per-batch draw seeds -- and main.py:80-94 is
applied to its output on the device.

Bit equality of the images is unattainable end to end (a random-init ReLU network's fp32 input gradient differs between
ANY two machines by ~1e-2, and the sign step amplifies that: DESIGN.md section 4), so the statement that can hold -- and
the one north_star asks for -- is equality of the attack success rate within the sampling error of 1000 images.

WHAT IS COUNTED.  Surrogates AND victims are seeded random-init networks (no checkpoints offline): every ASR here is a
statement about the two code paths on the same synthetic job, not about the published rows of the reference's README.  ASR
is counted two ways (gen_asr1000.py's docstring): "vs label" -- main.py:90 literally, against the label the attack used --
and "vs clean prediction" -- against the victim's own clean prediction.  With random-init victims the first is
information-free (a victim agrees with the surrogate's label on ~0 % of the clean images, so the rate is ~100 % on both
paths whatever the attack does): those rows are printed and NOT asserted unless the reference's rate is below 99 %, i.e.
unless the row can tell the paths apart (it would with trained weights).

WHAT IS ASSERTED, per informative row.  Both paths attack the SAME images, so the exact test is the paired one (McNemar):
with b / c the images only one of the two paths fools, b ~ Binomial(b + c, 1/2) under "same success rate".
  * paired: the exact two-sided binomial p-value of (b, c) must be >= P_MIN = 6.3e-5 (|z| = 4 for large counts).  Round 3
    used z <= 3.5 plus "|b - c| <= 3 when b + c < 9", and the latter tripped on the driver's box at (b, c) = (0, 4) -- a
    1-in-8 event under the null, on an information-free row.  The exact p-value needs no small-count patch.
  * unpaired: |ASR_gpu - ASR_ref| <= 4 * sqrt(2 p (1 - p) / n) -- the two-SAMPLE bound (two runs, each with its own
    sampling error).  The review's one-sample 3 * sqrt(p (1 - p) / n) is printed beside it and not asserted: the
    trajectories of the two paths decorrelate (78 % of the pixels differ), which images fall differs in ~2 p (1 - p) n
    cases, so that bound is only ~2.1 standard deviations of the difference and a correct implementation trips it in ~1 of
    30 comparisons (it was met by every comparison so far).
  False alarms: ~50 asserted rows per tier run x 2 x 6.3e-5 at most ~ 0.6 % per run that a correct implementation fails; a
  wrong sign, a stale momentum or a broken transform moves the rate by tens of points (|z| > 10).
Plus agreement of the FIRST-iteration gradient sign with the reference's (>= 99 %: the inputs of both surrogates are
identical there, whatever happens later).
"""
import os
import time

import numpy as np
import pytest
import torch

import transferattack_amd as ta
from conftest import GOLDEN_DIR, u8_images
from transferattack_amd import _hip, backbones
from transferattack_amd.utils import quantize_images, wrap_model

gpu = pytest.mark.gpu          # per test, not per module: the two ``gpu_long`` jobs at the end stay out of ``-m gpu``
DEV = "cuda"


def fixture(config):
    path = os.path.join(GOLDEN_DIR, "asr1000_%s.npz" % config)
    if not os.path.isfile(path):
        pytest.skip("tests/golden/asr1000_%s.npz has not been generated (oracle/gen_asr1000.py %s)" % (config, config))
    return np.load(path)


def product_attack(name, nets, fold_bn=False, channels_last=False):
    """the product's class ``name`` around the given seeded backbone(s) -- a list becomes an EnsembleModel (utils.py:82-105)"""
    from transferattack_amd.utils import EnsembleModel
    base = ta.load_attack_class(name)
    nets = nets if isinstance(nets, (list, tuple)) else [nets]

    def load_model(self, model_name):
        wrapped = []
        for backbone in nets:
            for p in backbone.parameters():
                p.requires_grad_(False)
            if fold_bn:
                backbones.fold_batchnorm(backbone)
            w = wrap_model(backbone.eval().to(DEV))
            from transferattack_amd.attack import takes_channels_last          # Attack.load_model's own rule (not Inception, not VGG)
            wrapped.append(w.to(memory_format=torch.channels_last) if channels_last and takes_channels_last(backbone) else w)
        return wrapped[0] if len(wrapped) == 1 else EnsembleModel(wrapped)

    return type("Gpu" + base.__name__, (base,), {"load_model": load_model})(model_name="injected")


def predictions(net, x, chunk=100):
    wrapped = wrap_model(net.eval().to(DEV))
    out = []
    with torch.no_grad():
        for i in range(0, len(x), chunk):
            out.append(wrapped(x[i:i + chunk].to(DEV)).argmax(1).cpu())
    wrapped.cpu()
    return torch.cat(out).numpy()


def run_config(config, name, arrangement, g, limit=None):
    """``limit``: only the first ``limit`` images of the fixture (whole reference batches; the per-batch seeds are the fixture's)"""
    n, batch, seed_base = int(g["n_images"]), int(g["batch"]), int(g["seed_base"])
    n = n if limit is None else min(n, limit)
    assert n % batch == 0 or n == int(g["n_images"])
    xu8 = u8_images(int(g["n_images"]), 224, int(g["seed_images"]))[:n]          # (the generator's stream: the first n of the set)
    x = xu8.float() / 255
    label = torch.from_numpy(g["label"].astype(np.int64))
    nets = [backbones.create(spec.split(":")[0], seed=int(spec.split(":")[1]), verbose=False) for spec in str(g["surrogate"]).split(",")]
    atk = product_attack(name, nets, **arrangement)
    first = []
    k = int(g["sign_images"])
    folded_plain = (name != "vmifgsm" and atk._can_fuse_update()
                    and atk._normalize_chain(x[:1].to(DEV).contiguous()) is not None)
    if name == "vmifgsm" or folded_plain:
        # the folded VMI loop never calls get_grad (gradient/vmifgsm.py): its own first-iteration gradient -- normalize_fwd,
        # the surrogate, normalize_bwd inside _forward_folded -- is taken through the loop's probe hook; so is the gradient of
        # the plain loop with the surrogate's Normalize folded into its two ends (attack.py::_forward_normalize_folded), which
        # patching get_grad would switch off
        def probe(it, grad):
            if it == 0 and not first:
                first.append(grad.detach().clone())
        atk.grad_probe = probe
    else:
        inner = type(atk).get_grad

        def get_grad(self, loss, delta, **kw):
            grad = inner(self, loss, delta, **kw)
            if not first:
                first.append(grad.detach().clone())
            return grad
        type(atk).get_grad = get_grad

    adv = np.empty((n, 224, 224, 3), np.uint8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range((n + batch - 1) // batch):
        lo, hi = b * batch, min((b + 1) * batch, n)
        torch.manual_seed(seed_base + b)                      # the DIM draws of batch b (host generator, reference's order)
        delta = atk(x[lo:hi], label[lo:hi])
        adv[lo:hi] = quantize_images(x[lo:hi], delta)         # utils.py:64 on the device
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    got = first[0][:k].cpu().numpy()
    ref_pos = np.unpackbits(g["sign_bits"])[:got.size].reshape(got.shape).astype(bool)
    agree = float(((got > 0) == ref_pos).mean())
    return x[:n], label.numpy()[:n], adv, agree, seconds


P_MIN = 2.2e-4          # two-sided tail of |z| = 3.7 (round 4 ran with |z| = 4; largest z observed over rounds 3-4: 2.93; ~1 % false alarms per 48-comparison tier)


def paired_p_value(b, c):
    """exact two-sided binomial (sign / McNemar) test of "b and c are draws of the same coin" """
    from scipy.stats import binomtest
    return 1.0 if b + c == 0 else float(binomtest(b, b + c, 0.5).pvalue)


def check_rates(config, tag, g, x, label, adv):
    n = len(label)
    x_adv = torch.from_numpy(adv).permute(0, 3, 1, 2).float() / 255
    rows = []
    for v, name in enumerate(g["victims"]):
        vname, vseed = str(name).split(":")
        net = backbones.create(vname, seed=int(vseed), verbose=False)
        clean_gpu, adv_gpu = predictions(net, x), predictions(net, x_adv)
        clean_ref, adv_ref = g["clean_pred"][v].astype(np.int64)[:n], g["adv_pred"][v].astype(np.int64)[:n]
        for kind, fooled_gpu, fooled_ref in (("vs label", adv_gpu != label, adv_ref != label),
                                             ("vs clean prediction", adv_gpu != clean_gpu, adv_ref != clean_ref)):
            p_gpu, p_ref = float(fooled_gpu.mean()), float(fooled_ref.mean())
            p = 0.5 * (p_gpu + p_ref)
            one_sample = max(3 * np.sqrt(p * (1 - p) / n), 3.0 / n)                  # the review's bound: printed
            two_sample = max(4 * np.sqrt(2 * p * (1 - p) / n), 4.0 / n)              # asserted
            only_gpu, only_ref = int((fooled_gpu & ~fooled_ref).sum()), int((~fooled_gpu & fooled_ref).sum())
            z = abs(only_gpu - only_ref) / np.sqrt(only_gpu + only_ref) if only_gpu + only_ref else 0.0
            informative = kind == "vs clean prediction" or p_ref < 0.99
            rows.append(dict(victim=str(name), kind=kind, p_ref=p_ref, p_gpu=p_gpu, one_sample=one_sample, two_sample=two_sample,
                             b=only_gpu, c=only_ref, z=float(z), p_value=paired_p_value(only_gpu, only_ref),
                             same_clean=float((clean_gpu == clean_ref).mean()), asserted=informative))
    print("\n%s [%s], %d images (seeded random-init surrogate and victims): attack success rate, reference (CPU) vs product "
          "(MI355X)" % (config, tag, n))
    for r in rows:
        print("  %-26s %-20s ref %6.2f %%   gpu %6.2f %%   |diff| %5.2f (two-sample 4-sigma bound %5.2f; one-sample 3-sigma "
              "%5.2f: %s)   discordant images %d / %d   paired z %.2f, exact p %.3g   clean predictions equal %.1f %%   %s"
              % (r["victim"], r["kind"], 100 * r["p_ref"], 100 * r["p_gpu"], 100 * abs(r["p_gpu"] - r["p_ref"]),
                 100 * r["two_sample"], 100 * r["one_sample"],
                 "met" if abs(r["p_gpu"] - r["p_ref"]) <= r["one_sample"] else "exceeded", r["b"], r["c"], r["z"], r["p_value"],
                 100 * r["same_clean"], "asserted" if r["asserted"] else "not asserted: information-free for random-init victims"))
    assert any(r["asserted"] and 0.02 < r["p_ref"] < 0.98 for r in rows), "no victim can tell the two paths apart"
    for r in rows:
        if not r["asserted"]:
            continue
        assert r["p_value"] >= P_MIN, "%s %s: ASR %.2f %% on the GPU vs %.2f %% by the reference (discordant %d / %d, p = %.2g)" % (
            r["victim"], r["kind"], 100 * r["p_gpu"], 100 * r["p_ref"], r["b"], r["c"], r["p_value"])
        assert abs(r["p_gpu"] - r["p_ref"]) <= r["two_sample"], "%s %s: ASR %.2f %% vs %.2f %% (two-sample 4-sigma bound %.2f)" % (
            r["victim"], r["kind"], 100 * r["p_gpu"], 100 * r["p_ref"], 100 * r["two_sample"])
    return rows


@gpu
@pytest.mark.parametrize("tag,arrangement", [("reference-literal surrogate", dict()),
                                             ("folded BatchNorm + NHWC + fused glue + CK epilogues, as bench.py", dict(fold_bn=True, channels_last=True))])
def test_asr1000_mifgsm_resnet50(tag, arrangement, monkeypatch):
    """BASELINE.json configs[1]: MI-FGSM, ResNet-50, eps 16/255, alpha 1.6/255, K = 10, the 1000-image set; the second case in the
    arrangement bench.py times, glue passes in the convolutions' epilogues (libta_ck.so) included"""
    from transferattack_amd import _ck
    g = fixture("mifgsm")
    before = dict(_hip.stats)
    ck_before = _ck.stats["fused_launches"]
    monkeypatch.setenv("TA_CK_EPILOGUE", "1" if arrangement else "0")
    x, label, adv, agree, seconds = run_config("configs[1]", "mifgsm", arrangement, g)
    if arrangement:
        assert _ck.stats["fused_launches"] > ck_before, "no convolution ran with its glue as epilogue"
        print("convolution sites tuned %d, on composable_kernel with the glue as epilogue %d" % (_ck.stats["tuned_sites"], _ck.stats["sites_on_ck"]))
    print("\nconfigs[1] [%s]: %d images in %.1f s (%.0f images/s incl. upload, quantise, download); first-iteration "
          "gradient sign agreement with the reference %.3f %%" % (tag, len(label), seconds, len(label) / seconds, 100 * agree))
    assert _hip.stats["std_form_launches"] > before["std_form_launches"], "the loop did not fold the Normalize into its ends"
    if arrangement:         # bench.py's arrangement: the stem kernel leaves the |gy / std| sums -- no pass re-reads the gradient
        assert _hip.stats["k1_passes"] == before["k1_passes"], "the fused update re-read the gradient"
    assert agree >= 0.99
    check_rates("configs[1] MI-FGSM / ResNet-50", tag, g, x, label, adv)


@gpu
@pytest.mark.parametrize("tag,arrangement", [("reference-literal surrogate", dict()),
                                             ("folded BatchNorm + NHWC + fused glue, as bench.py", dict(fold_bn=True, channels_last=True))])
def test_asr1000_dts_resnet50(tag, arrangement):
    """BASELINE.json configs[2]: DIM + TIM + SIM (5 scale copies), ResNet-50, K = 10, the 1000-image set"""
    g = fixture("dts")
    x, label, adv, agree, seconds = run_config("configs[2]", "dts", arrangement, g)
    print("\nconfigs[2] [%s]: %d images in %.1f s (%.0f images/s); first-iteration gradient sign agreement with the "
          "reference %.3f %%" % (tag, len(label), seconds, len(label) / seconds, 100 * agree))
    assert agree >= 0.99
    check_rates("configs[2] DTS / ResNet-50", tag, g, x, label, adv)


# The two slowest jobs of this file (3.2 and 7 images/s on one device) run on the FIRST 256 images -- 8 reference batches -- in
# the ``-m gpu`` tier, whose whole run has to fit the driver's step limit, and on all 1000 behind the second marker
# ``gpu_long`` (``pytest -m gpu_long``; tools/gpu_check.sh ``asrlong`` / ``tests`` run it).  Same assertions either way; with 256
# images the bounds are wider by a factor of two, a wrong sign or a stale momentum still moves the rate by tens of points.
SHORT = 256


def _ens_four_members(limit):
    g = fixture("ens")
    x, label, adv, agree, seconds = run_config("configs[4]", "ens", dict(fold_bn=True, channels_last=True), g, limit)
    print("\nconfigs[4]: %d images in %.1f s (%.0f images/s); first-iteration gradient sign agreement with the reference "
          "%.3f %%" % (len(label), seconds, len(label) / seconds, 100 * agree))
    assert agree >= 0.99
    check_rates("configs[4] ensemble MI-FGSM / RN50 + VGG-16 + Inc-v3 + ViT-B/16", "the bench arrangement", g, x, label, adv)


def _vmifgsm_vit(limit):
    g = fixture("vmifgsm")
    x, label, adv, agree, seconds = run_config("configs[3]", "vmifgsm", dict(), g, limit)
    print("\nconfigs[3]: %d images in %.1f s (%.1f images/s); first-iteration gradient sign agreement with the reference "
          "%.3f %%" % (len(label), seconds, len(label) / seconds, 100 * agree))
    assert agree >= 0.99
    check_rates("configs[3] VMI-FGSM / ViT-B/16", "reference-literal surrogate", g, x, label, adv)


@gpu
def test_asr_ens_four_members():
    """BASELINE.json configs[4] on one device: ensemble MI-FGSM over ResNet-50 + VGG-16 + Inception-v3 + ViT-B/16 (logit mean,
    utils.py:94-101), the first 256 images of the 1000-image set (the reference needs ~4 s of CPU time per image here), in the
    arrangement ``bench.py --attack ens`` runs: BatchNorm folded, ResNet-50 through the fused glue with ReLU pass bits, NHWC for
    ResNet / ViT only (attack.takes_channels_last), the 299-pixel member behind the resize + Normalize kernel, every member on
    its own HIP stream.  (The reference-literal arrangement of these members is what test_config5_ensemble_replay pins.)"""
    _ens_four_members(SHORT)


@gpu
def test_asr_vmifgsm_vit():
    """BASELINE.json configs[3] on one device: VMI-FGSM on ViT-B/16, 20 neighbours, the first 256 images of the 1000-image set
    (the whole fixture: 32 reference batches = 6720 surrogate evaluations of 32 images, ~6.5 h of reference CPU time).  The
    neighbours come from different generators on the two paths (torch's CPU generator in the reference run, the in-kernel
    Philox stream here) -- as they would between any two runs of the reference itself, which seeds nothing."""
    _vmifgsm_vit(SHORT)


@pytest.mark.gpu_long
def test_asr1000_ens_four_members():
    """test_asr_ens_four_members on all 1000 images (155 s on MI355X)"""
    _ens_four_members(None)


@pytest.mark.gpu_long
def test_asr1000_vmifgsm_vit():
    """test_asr_vmifgsm_vit on all 1000 images (330 s on MI355X)"""
    _vmifgsm_vit(None)
