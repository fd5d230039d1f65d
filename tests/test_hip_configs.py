"""GPU (-m gpu): parity at the sizes and on the surrogates BASELINE.json's configs name (224 x 224, ResNet-50 / ViT-B/16 /
the four-member ensemble) -- the toy-CNN loop tests of test_hip_attacks.py repeated where the numbers are quoted.

Tier 1 -- identical gradient source => identical bytes.  The oracle (pinned bit for bit to the REAL reference's outputs
for these very configurations by tests/test_oracle_golden.py: tests/golden/config{2,3,4,5}_*.npz, written by
oracle/gen_golden.py) runs the configuration on this host's CPU and records every gradient the loop consumes; the
product's attack class then runs on the GPU -- HIP transforms, surrogate forward/backward, HIP update -- with each
gradient replaced by the recorded one.  The attack runs in its DEFAULT loop form: for the plain attacks (MI-FGSM: configs[1])
that is attack.py::_forward_normalize_folded (ta_normalize_adv_fwd -> backbone -> ta_mi_update_std), which never
materialises d(loss)/d(delta); there the recorded gradient at the BACKBONE'S INPUT is injected (``Attack.grad_inject``) and the
update kernel forms gy / std itself -- ``std_form_launches == K`` asserted; the attacks that override a hook (DTS, VMI, the
ensemble) run the hook loop, where ``get_grad`` is replaced.  By induction the iterates coincide, so
  * the final perturbation and the uint8 images must equal the oracle's BIT FOR BIT (and the reference's golden bytes,
    whenever this host's CPU reproduces them -- oneDNN's summation order depends on the CPU model);
  * the GPU's own fp32 gradient is compared with the reference's at the same point, iteration by iteration (reported;
    sign agreement asserted).
Tier 2 -- the arrangement bench.py measures (eval-mode BatchNorm folded into the convolutions + NHWC) computes the same
function as the reference-literal one: logits and input-gradients of both, on the device, against an fp64 ground truth.
"""
import os

import numpy as np
import pytest
import torch

import fgsm_oracle as O
import transferattack_amd as ta
from conftest import u8_images
from transferattack_amd import _hip, backbones
from transferattack_amd.utils import EnsembleModel, quantize_images, wrap_model

pytestmark = pytest.mark.gpu
EPS, ALPHA = 16 / 255, 1.6 / 255
DEV = "cuda"
ENS_MEMBERS = ("resnet50", "vgg16", "inception_v3", "vit_base_patch16_224")
FOLDED_LOOP = ("fgsm", "ifgsm", "mifgsm")        # attacks whose default loop on a 224-pixel surrogate is _forward_normalize_folded


def t(a):
    return torch.from_numpy(np.asarray(a))


def product_attack(name, models, **kw):
    """the product's class ``name`` around the given (CPU-built, seeded) backbones moved to the device"""
    base = ta.load_attack_class(name)

    def load_model(self, model_name):
        wrapped = [wrap_model(m.eval().to(DEV)) for m in models]
        return wrapped[0] if len(wrapped) == 1 else EnsembleModel(wrapped)

    atk = type("Gpu" + base.__name__, (base,), {"load_model": load_model})(model_name="injected", **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)      # the reference's CPU draws
    return atk


def host_sum_lanes():
    """SIMD width (8 = AVX2, 16 = AVX-512) of the cascade order in which THIS host's ATen adds ``grad.abs().mean()``
    (attack.py:128), found by comparing torch's own sum with the two restatements of oracle/ta_oracle.c"""
    import c_oracle as C
    gen = torch.Generator().manual_seed(123)
    rows = torch.randn(3, 150528, generator=gen).abs()
    got = rows.sum(dim=1).numpy()
    for lanes in (8, 16):
        if all(got[i] == C.aten_row_sum(rows[i].numpy(), lanes) for i in range(rows.shape[0])):
            return lanes
    return None


def replay(monkeypatch, name, model_names, n, golden_arrays, key, oracle_kw=None, atk_kw=None, draw_seed=None,
           max_flip_first=0.02):
    """-> dict of statistics; asserts the bit-exact claims of tier 1 (module docstring)"""
    oracle_kw, atk_kw = oracle_kw or {}, atk_kw or {}
    x = u8_images(n, 224, int(golden_arrays["seed_images"])).float() / 255
    label = t(golden_arrays["label"])
    seed_w = int(golden_arrays["seed_weights"])
    cpu_models = [backbones.create(m, seed=seed_w, verbose=False) for m in model_names]
    if draw_seed is not None:
        torch.manual_seed(draw_seed)
    trace = []
    delta_ref = O.run_attack(name, cpu_models if len(cpu_models) > 1 else cpu_models[0], x, label, trace=trace, **oracle_kw)
    ref_grads = [g for rec in trace for g in rec["grads"]]
    ref_gy = [g for rec in trace for g in rec.get("grads_y", [])]
    folded_loop = name in FOLDED_LOOP
    u8_ref = O.quantize_u8(x + delta_ref)
    host_matches_golden = np.array_equal(u8_ref, golden_arrays[key])
    lanes = host_sum_lanes()
    s = None
    for mode in ("kernel-order", "reference-order"):
        # kernel-order: sum|g| in the kernels' own fixed order -> g/mean|g| can differ from ATen's in the last bit, so a
        #   momentum within rounding of zero may take the other sign: such pixels, and only they, may differ (<= 1e-5 of
        #   all elements; the gradients are replayed, so a flip cannot spread).
        # reference-order (TA_ATEN_SUM_LANES = this host's ATen SIMD width): the same expression tree as the oracle's
        #   CPU sum -> every iterate, the perturbation and the uint8 images BIT FOR BIT.
        if mode == "reference-order":
            if lanes is None:
                print("this host's ATen sum order is neither the 8- nor the 16-lane cascade: reference-order pass skipped")
                continue
            monkeypatch.setenv("TA_ATEN_SUM_LANES", str(lanes))
        gpu_models = [backbones.create(m, seed=seed_w, verbose=False) for m in model_names]
        atk = product_attack(name, gpu_models, **atk_kw)
        stats, it = [], [0]
        orig_get_grad = type(atk).get_grad

        def compare(gpu):
            ref = ref_grads[it[0]]
            diff = (gpu - ref).abs()
            stats.append((float((gpu - ref).norm() / ref.norm()), float((diff <= 1e-5 * ref.abs().max()).float().mean()),
                          float((torch.sign(gpu) != torch.sign(ref)).float().mean())))
            it[0] += 1
            return ref

        def get_grad(self, loss, delta, **kw):
            return compare(orig_get_grad(self, loss, delta, **kw).cpu()).to(DEV)

        def inject(iteration, gy):
            # the default loop of this attack never forms d(loss)/d(delta): it hands gy, the backbone's own input gradient, to
            # ta_mi_update_std, which divides by std[c] inline.  Replayed: the reference's gy of this iteration, whose
            # Normalize backward (utils.py:76: gy / std) IS the gradient the reference's get_grad returned -- checked here
            std = atk.model[0].normalize.std.reshape(1, -1, 1, 1)
            compare((gy / std).cpu())
            want = ref_gy[iteration]
            assert torch.equal(want / std.cpu(), ref_grads[iteration]), "recorded gy / std is not the recorded gradient"
            return want.to(DEV)

        if folded_loop:
            assert len(ref_gy) == len(ref_grads), "the oracle recorded no gradient at the backbone's input"
            atk.grad_inject = inject
        else:
            type(atk).get_grad = get_grad
        if draw_seed is not None:
            torch.manual_seed(draw_seed)
        launches = _hip.stats["std_form_launches"]
        delta = atk(x, label)
        monkeypatch.delenv("TA_ATEN_SUM_LANES", raising=False)
        assert it[0] == len(ref_grads), "the loop asked for %d gradients, the reference for %d" % (it[0], len(ref_grads))
        if folded_loop:           # it WAS the default loop form (the one bench.py times), one std-form update per iteration
            assert _hip.stats["std_form_launches"] - launches == len(ref_grads) == atk.epoch
        s = np.array(stats)
        u8 = quantize_images(x, delta)
        d_bad = float((delta.cpu() != delta_ref).float().mean())
        u8_bad = float((u8 != u8_ref).mean())
        print("%s on %s, %d images [%s]: oracle on this host reproduces the reference's golden uint8: %s (mismatch %.4f%%); "
              "GPU vs reference input-gradient over %d evaluations: rel-L2 first %.2e / worst %.2e, within 1e-5*max|g| first "
              "%.2f%% / worst %.2f%%, sign flips first %.3f%% / worst %.3f%%; delta elements differing from the oracle's "
              "%.2e, uint8 %.2e"
              % (name, "+".join(model_names), n, mode, host_matches_golden,
                 100 * float((u8_ref != golden_arrays[key]).mean()), len(stats), s[0, 0], s[:, 0].max(), 100 * s[0, 1],
                 100 * s[:, 1].min(), 100 * s[0, 2], 100 * s[:, 2].max(), d_bad, u8_bad))
        assert s[0, 2] <= max_flip_first, "first gradient already disagrees in sign: the inputs of the surrogate differ"
        if mode == "kernel-order":
            assert d_bad <= 1e-5 and u8_bad <= 1e-5, "more than rounding-of-zero momentum flips"
        else:
            assert torch.equal(delta.cpu(), delta_ref), "perturbation differs from the oracle's with identical gradients"
            assert np.array_equal(u8, u8_ref)
            if host_matches_golden:
                assert np.array_equal(u8, golden_arrays[key])               # ... and the reference's own bytes
    return dict(host_matches_golden=host_matches_golden, stats=s)


def test_config2_mifgsm_resnet50_replay(golden, monkeypatch):
    """BASELINE.json configs[1] in miniature: MI-FGSM, ResNet-50, 224 x 224, K = 10, 4 images"""
    before = dict(_hip.stats)
    replay(monkeypatch, "mifgsm", ["resnet50"], 4, golden("config2_mifgsm_resnet50_n4"), "adv_u8")
    assert _hip.stats["partials_reused"] == before["partials_reused"]      # replayed gradients are fresh tensors: a sum-only
    assert _hip.stats["k1_passes"] - before["k1_passes"] in (10, 20)       # pass (ta_abs_sum_partials_std) each time: K per sum order


def test_config3_dts_resnet50_replay(golden, monkeypatch):
    """configs[2] in miniature: DTS = DIM o SIM on the input (5 copies, one geometry per iteration), TIM on the gradient,
    ResNet-50, K = 10, 2 images -> 10-image surrogate batches"""
    g = golden("config3_dts_resnet50_n2")
    replay(monkeypatch, "dts", ["resnet50"], 2, g, "adv_u8", draw_seed=int(g["seed_draws"]))


@pytest.mark.parametrize("tag,model,kw", [("vit", "vit_base_patch16_224", dict(num_neighbor=4, epoch=3)),
                                          ("resnet18", "resnet18", dict(num_neighbor=20, epoch=3))])
def test_config4_vmifgsm_replay(golden, monkeypatch, tag, model, kw):
    """configs[3] in miniature: VMI-FGSM (gradient/vmifgsm.py:42-97) at 224 x 224 on ViT-B/16 (4 variance samples, K = 3)
    and on ResNet-18 with the full 20 samples (K = 3): neighbour draws, gradient accumulation, variance, momentum on
    grad + variance, update -- every gradient of the loop (1 + num_neighbor per iteration) replayed"""
    g = golden("config4_vmifgsm_n2")
    replay(monkeypatch, "vmifgsm", [model], 2, g, "adv_u8_" + tag, oracle_kw=kw, atk_kw=kw, draw_seed=int(g["seed_draws"]),
           max_flip_first=0.05)


def test_config5_ensemble_replay(golden, monkeypatch):
    """configs[4] in miniature on ONE device: ResNet-50 + VGG-16 + Inception-v3 (299-pixel branch of wrap_model) +
    ViT-B/16 through EnsembleModel (logit mean; the members' input gradients added by ta_sum_members), K = 3, 2 images"""
    out = replay(monkeypatch, "ens", list(ENS_MEMBERS), 2, golden("config5_ens4_n2"), "adv_u8", oracle_kw=dict(epoch=3),
                 atk_kw=dict(epoch=3), max_flip_first=0.05)
    assert out["stats"].shape[0] == 3


@pytest.mark.parametrize("name,model", [("mifgsm", "resnet50"), ("dts", "resnet50")])
def test_end_to_end_at_config_size(golden, name, model):
    """The whole loop on the device against the reference's golden bytes, for the record: invariants asserted, the
    mismatch REPORTED.  A seeded random-init ResNet-50 amplifies a last-bit difference of one gradient into a different
    trajectory (two CPUs already disagree), so no bound on the mismatch is claimed here -- the claims that
    hold are tier 1 above and the per-gradient accuracy of test_hip_attacks.py::test_gradient_accuracy_vs_fp64."""
    g = golden("config2_mifgsm_resnet50_n4" if name == "mifgsm" else "config3_dts_resnet50_n2")
    n = len(g["label"])
    x = u8_images(n, 224, int(g["seed_images"])).float() / 255
    atk = product_attack(name, [backbones.create(model, seed=int(g["seed_weights"]), verbose=False)])
    before = dict(_hip.stats)
    torch.manual_seed(int(g["seed_draws"]) if "seed_draws" in g.files else 0)
    delta = atk(x, t(g["label"])).cpu()
    if name == "mifgsm":      # the plain loop folds the Normalize into its ends; this surrogate's backward (module path) is
        # MIOpen's, so a sum-only pass over gy precedes each update (4 + 21 instead of 8 + 25 B/element)
        assert _hip.stats["std_form_launches"] == before["std_form_launches"] + 10
    else:
        assert _hip.stats["partials_reused"] == before["partials_reused"] + 10, "the fused update re-read the gradient"
    assert float(delta.abs().max()) <= EPS + 1e-7
    adv = x + delta
    assert float(adv.min()) >= 0.0 and float(adv.max()) <= 1.0 + 1e-7
    u8 = quantize_images(x, delta)
    print("%s on %s end to end: uint8 mismatch vs the reference's golden bytes %.2f%%, identical images %d of %d"
          % (name, model, 100 * float((u8 != g["adv_u8"]).mean()),
             int((u8 == g["adv_u8"]).reshape(n, -1).all(1).sum()), n))


# ------------------------------------------------------------------------------------------------ tier 2
def _truth(name, x, label):
    """logits and d(loss)/dx of the seeded surrogate in fp64 on the CPU (preprocessing included)"""
    m = backbones.create(name, seed=0, verbose=False).double()
    cfg = O.preprocess_cfg(m)
    xin = x.double().requires_grad_(True)
    logits = m(O.preprocess(xin, cfg[0], [float(v) for v in cfg[1]], [float(v) for v in cfg[2]]))
    grad = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, label), xin)[0]
    return logits.detach(), grad


@pytest.mark.parametrize("name", ["resnet18", "resnet50", "mobilenet_v2", "inception_v3", "vgg16", "vgg16_nhwc", "vit_base_patch16_224"])
def test_fold_bn_channels_last_is_the_same_surrogate(monkeypatch, name):
    """bench.py's arrangement (TA_FOLD_BN=1 TA_CHANNELS_LAST=1, through Attack.load_model exactly as bench.py builds it)
    against the reference-literal one (separate BatchNorm, NCHW): both on the device in fp32, both against the fp64
    truth.  Asserted: logits as accurate as the literal arrangement's (<= 4x its relative L2 error, or 1e-5), the same sign on
    >= 99% of the gradient -- everything the attack uses -- and an input-gradient error within 4x the literal one's OR below an
    absolute floor of 3e-2.  The floor is there because the gradient error of a seeded random-init surrogate is a run-to-run
    noisy quantity on the device (MIOpen's algorithm choice per fresh find-db, atomic accumulation in backward-data): for
    VGG-16 in NHWC the ratio bench / literal was 1.36, 2.06, 2.88, 3.02, 3.25 in rounds 2-3 and 4.002 on the round-3 driver
    box, where a bare 4x bound stopped the -x run.  The finding under that noise -- VGG-16's input gradient is 2-4x further
    from the fp64 truth in NHWC than in NCHW (4.4e-3 -> <= 1.75e-2) -- is acted on, not hidden: ``Attack.load_model`` no longer
    puts the VGGs in NHWC (attack.py; ``TA_VGG_CHANNELS_LAST=1`` restores it), and the ``vgg16_nhwc`` case below keeps
    measuring and printing the pair so the number stays on record."""
    n = 2
    monkeypatch.setenv("TA_VGG_CHANNELS_LAST", "1" if name.endswith("_nhwc") else "0")
    name = name.replace("_nhwc", "")
    x = u8_images(n, 224, 5).float() / 255
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(6))
    logits64, grad64 = _truth(name, x, label)
    got = {}
    for tag, fold, nhwc in (("literal", "0", "0"), ("bench", "1", "1")):
        monkeypatch.setenv("TA_FOLD_BN", fold)
        monkeypatch.setenv("TA_CHANNELS_LAST", nhwc)
        atk = ta.load_attack_class("mifgsm")(model_name=name)
        if tag == "bench" and any(isinstance(m, torch.nn.BatchNorm2d) for m in backbones.create(name, verbose=False).modules()):
            assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in atk.model.modules()), "BatchNorm left unfolded"
        xd = x.to(DEV).requires_grad_(True)
        logits = atk.model(xd)
        grad = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, label.to(DEV)), xd)[0]
        got[tag] = (logits.detach().cpu().double(), grad.cpu().double())
    rel = lambda a, b: float((a - b).norm() / b.norm())       # noqa: E731
    e_lit = (rel(got["literal"][0], logits64), rel(got["literal"][1], grad64))
    e_bench = (rel(got["bench"][0], logits64), rel(got["bench"][1], grad64))
    flips = float((torch.sign(got["bench"][1]) != torch.sign(got["literal"][1])).float().mean())
    print("%s: rel-L2 error vs fp64 truth (logits, input-gradient): reference-literal %.2e %.2e; folded-BN + NHWC %.2e %.2e; "
          "gradient sign flips between the two %.3f%%" % (name, e_lit[0], e_lit[1], e_bench[0], e_bench[1], 100 * flips))
    # per-surrogate floors from the spread measured over rounds 2-4 on MI355X (~1.5x the worst figure of either
    # arrangement): a surrogate several times worse than it has ever been fails; 3e-2 stays for the VGG-16 NHWC pair only
    floor = {"resnet18": 1e-2, "resnet50": 2.5e-2, "mobilenet_v2": 2e-2, "inception_v3": 2.8e-2, "vgg16": 1e-2,
             "vit_base_patch16_224": 1e-5}[name]
    if os.environ.get("TA_VGG_CHANNELS_LAST") == "1":
        floor = 3e-2
    assert e_bench[0] <= max(4 * e_lit[0], 1e-5) and e_bench[1] <= max(4 * e_lit[1], floor)
    assert flips <= 0.01


@pytest.mark.parametrize("name,nhwc,batch", [("resnet50", "1", 4), ("resnet50", "0", 2), ("resnet18", "1", 4)])
def test_fused_glue_is_the_same_surrogate(monkeypatch, name, nhwc, batch):
    """backbones/fused.py (the surrogate's bias / ReLU / residual / threshold passes fused, csrc/glue.hip; the stem's input
    gradient on csrc/stem.hip) against the plain module path of the same folded surrogate, both on the device, both against
    the fp64 truth.  On the CPU tier the two are EQUAL bit for bit (tests/test_fused_backbone.py); on the device MIOpen may
    pick other algorithms for a convolution called without its bias, and a seeded random-init ResNet amplifies any
    rounding-level change of an activation to ~1e-2 of the input gradient (two CPUs differ as much) --
    so the claim checked here is the one of test_fold_bn_channels_last_is_the_same_surrogate: as accurate as the module
    path, same gradient sign on >= 99 %, logits equal to rounding.  "As accurate" has a floor of 1e-2: the module path's OWN
    error on the well-conditioned ResNet-18 moves between 9e-7 and 3e-3 from run to run with MIOpen's algorithm choice (r3c /
    r3d: Winograd variants for the 3 x 3 convolutions once the find-db knows them), and neither path is run-to-run
    deterministic on the device (atomic accumulation in backward-data and max-pool backward)."""
    x = u8_images(batch, 224, 5).float() / 255
    label_cpu = torch.randint(0, 1000, (batch,), generator=torch.Generator().manual_seed(6))
    label = label_cpu.to(DEV)
    logits64, grad64 = _truth(name, x, label_cpu)
    monkeypatch.setenv("TA_FOLD_BN", "1")
    monkeypatch.setenv("TA_CHANNELS_LAST", nhwc)
    atk = ta.load_attack_class("mifgsm")(model_name=name)
    got = {}
    from transferattack_amd import _ck
    ck_before = _ck.stats["fused_launches"]
    for tag, flag, stem, ck in (("module", "0", "1", "0"), ("module again", "0", "1", "0"), ("fused", "1", "1", "0"),
                                ("fused again", "1", "1", "0"), ("fused, MIOpen stem", "1", "0", "0"), ("fused, CK epilogues", "1", "1", "1")):
        monkeypatch.setenv("TA_FUSED_GLUE", flag)
        monkeypatch.setenv("TA_STEM_KERNEL", stem)
        monkeypatch.setenv("TA_CK_EPILOGUE", ck)
        xd = x.to(DEV).requires_grad_(True)
        logits = atk.model(xd)
        grad = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, label), xd)[0]
        got[tag] = (logits.detach().cpu().double(), grad.cpu().double())
    rel = lambda a, b: float((a - b).norm() / b.norm())       # noqa: E731
    e_mod = (rel(got["module"][0], logits64), rel(got["module"][1], grad64))
    e_fus = (rel(got["fused"][0], logits64), rel(got["fused"][1], grad64))
    flips = float((torch.sign(got["fused"][1]) != torch.sign(got["module"][1])).float().mean())
    e_mio = (rel(got["fused, MIOpen stem"][0], logits64), rel(got["fused, MIOpen stem"][1], grad64))
    print("%s nhwc=%s: rel-L2 error vs fp64 truth (logits, input-gradient): module path %.2e %.2e; fused glue %.2e %.2e (with "
          "MIOpen's stem backward %.2e %.2e); fused vs module: logits rel %.1e, gradient rel %.1e, sign flips %.3f%%; run-to-run "
          "gradient equal: module %s, fused %s"
          % (name, nhwc, e_mod[0], e_mod[1], e_fus[0], e_fus[1], e_mio[0], e_mio[1], rel(got["fused"][0], got["module"][0]),
             rel(got["fused"][1], got["module"][1]), 100 * flips, torch.equal(got["module"][1], got["module again"][1]),
             torch.equal(got["fused"][1], got["fused again"][1])))
    assert rel(got["fused"][0], got["module"][0]) <= 1e-5
    assert e_fus[0] <= max(4 * e_mod[0], 1e-5) and e_fus[1] <= max(4 * e_mod[1], 1e-2)
    assert flips <= 0.01
    # the same surrogate with the glue passes in the convolutions' epilogues (libta_ck.so; NHWC only): same claims
    e_ck = (rel(got["fused, CK epilogues"][0], logits64), rel(got["fused, CK epilogues"][1], grad64))
    flips_ck = float((torch.sign(got["fused, CK epilogues"][1]) != torch.sign(got["module"][1])).float().mean())
    print("%s nhwc=%s, CK epilogues: rel-L2 error vs fp64 (logits, input-gradient) %.2e %.2e; vs module path: logits rel %.1e, sign "
          "flips %.3f%%; fused launches %d, sites on CK %d of %d tuned" % (name, nhwc, e_ck[0], e_ck[1], rel(got["fused, CK epilogues"][0], got["module"][0]),
                                                                           100 * flips_ck, _ck.stats["fused_launches"] - ck_before,
                                                                           _ck.stats["sites_on_ck"], _ck.stats["tuned_sites"]))
    # (other convolution kernels than MIOpen's pick, other accumulation order: logits equal to fp32 rounding through 50 layers,
    # 1.3e-5 measured; what bounds the path is its distance from the fp64 truth, next line)
    assert rel(got["fused, CK epilogues"][0], got["module"][0]) <= 5e-5
    assert e_ck[0] <= max(4 * e_mod[0], 1e-5) and e_ck[1] <= max(4 * e_mod[1], 1e-2) and flips_ck <= 0.01
    if nhwc == "1":
        assert _ck.stats["fused_launches"] > ck_before, "TA_CK_EPILOGUE=1 launched no fused convolution"


@pytest.mark.parametrize("n,oh,ow", [(4, 112, 112), (1, 9, 37), (2, 150, 150)])
def test_stem_input_grad_kernel_on_device(n, oh, ow):
    """csrc/stem.hip on MI355X against MIOpen's backward-data and an fp64 evaluation of the same convolution backward: at
    least as accurate as MIOpen (<= 4 x its error, or 2e-6 of max|dx|) -- the 224 / 299-pixel stems and a ragged shape"""
    gen = torch.Generator().manual_seed(7)
    w = torch.randn(64, 3, 7, 7, generator=gen) * 0.05
    dy = torch.randn(n, 64, oh, ow, generator=gen)
    spec = ([2, 2], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False])
    truth = torch.ops.aten.convolution_backward(dy.double(), torch.empty(n, 3, 2 * oh, 2 * ow).double(), w.double(), None, *spec)[0]
    dyd = dy.to(DEV).contiguous(memory_format=torch.channels_last)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    xd = torch.empty(n, 3, 2 * oh, 2 * ow, device=DEV)
    ref = torch.ops.aten.convolution_backward(dyd, xd, wd, None, *spec)[0].cpu().double()
    got = _hip.stem7s2_input_grad(dyd, _hip.stem7s2_prepare(wd), torch.full_like(xd, float("nan"))).cpu().double()
    again = _hip.stem7s2_input_grad(dyd, _hip.stem7s2_prepare(wd), torch.full_like(xd, float("nan"))).cpu().double()
    scale = float(truth.abs().max())
    e_got, e_ref = float((got - truth).abs().max()) / scale, float((ref - truth).abs().max()) / scale
    bad = (got - truth).abs() > 1e-4 * scale
    print("stem input gradient [%d, 64, %d, %d]: max error / max|dx| vs fp64: csrc/stem.hip %.2e, MIOpen %.2e; elements off by more "
          "than 1e-4: %d of %d; run-to-run equal: %s" % (n, oh, ow, e_got, e_ref, int(bad.sum()), bad.numel(), torch.equal(got, again)))
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        print("  first offenders (n, c, y, x):", idx)
    assert not torch.isnan(got).any() and torch.equal(got, again)
    assert e_got <= max(4 * e_ref, 2e-6)
    # dy in NCHW memory (the module path of the reference-literal arrangement): the same bits
    nchw = _hip.stem7s2_input_grad(dy.to(DEV).contiguous(), _hip.stem7s2_prepare(wd), torch.full_like(xd, float("nan"))).cpu().double()
    assert torch.equal(nchw, got)


@pytest.mark.parametrize("shape", [(4, 64, 112, 112), (2, 8, 6, 14), (1, 128, 2, 2)])
def test_stem_pool_pair_on_device(shape):
    """ta_maxpool3s2_fwd / ta_maxpool3s2_bwd_relu on MI355X against ATen's device kernels: the pooled values, WHICH element wins
    every window (ties at the ReLU's zeros, ties between equal positives, NaN) and the activation's pass bits are EQUAL to
    max_pool2d_with_indices; the gradient equals the generic gather (ta_maxpool_bwd_relu on ATen's int64 indices) bit for bit
    and ATen's atomic-add backward to summation order"""
    gen = torch.Generator().manual_seed(sum(shape))
    cl = torch.channels_last
    n, c, h, w = shape
    y = (torch.randn(shape, generator=gen).clamp_min(0) * 4).round().div(4)
    y[0, :, 0, 0] = float("nan")
    y = y.to(DEV).contiguous(memory_format=cl)
    want, idx = torch.nn.functional.max_pool2d(y, 3, 2, 1, return_indices=True)
    pooled, arg, bits = _hip.maxpool3s2_fwd(y)
    assert torch.equal(torch.nan_to_num(pooled, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
    kh, kw = arg.long() // 3, arg.long() % 3
    ii = torch.arange(h // 2, device=DEV).view(1, 1, -1, 1)
    jj = torch.arange(w // 2, device=DEV).view(1, 1, 1, -1)
    assert torch.equal((2 * ii - 1 + kh) * w + (2 * jj - 1 + kw), idx)
    flat = (~(y <= 0)).permute(0, 2, 3, 1).reshape(-1, 8).to(torch.uint8)
    assert torch.equal(bits, (flat << torch.arange(8, dtype=torch.uint8, device=DEV)).sum(1).to(torch.uint8))
    ga = torch.randn(want.shape, generator=gen).to(DEV).contiguous(memory_format=cl)
    gb = torch.randn(want.shape, generator=gen).to(DEV).contiguous(memory_format=cl)
    ref = torch.ops.aten.max_pool2d_with_indices_backward(ga + gb, y, [3, 3], [2, 2], [1, 1], [1, 1], False, idx)
    ref = torch.ops.aten.threshold_backward(ref, y, 0)
    generic = _hip.maxpool_bwd_relu(ga, idx.contiguous(memory_format=cl), y, torch.full_like(y, float("nan")), 3, 2, 1, gb=gb)
    got = _hip.maxpool3s2_bwd_relu(ga, arg, bits, torch.full_like(y, float("nan")), gb=gb)
    assert not torch.isnan(got).any() and torch.equal(got, generic)
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape,k,s,p", [((4, 64, 112, 112), 3, 2, 1), ((2, 8, 9, 13), 3, 2, 1), ((2, 16, 12, 12), 2, 2, 0)])
def test_maxpool_backward_relu_kernel_on_device(shape, k, s, p):
    """ta_maxpool_bwd_relu on MI355X against the three ATen passes it replaces (junction add, max_pool2d_with_indices_backward
    -- atomic adds, so only equal to fp32 summation order -- and threshold_backward); deterministic itself"""
    gen = torch.Generator().manual_seed(11)
    cl = torch.channels_last
    y = torch.randn(shape, generator=gen).clamp_min(0).to(DEV).contiguous(memory_format=cl)
    pooled, idx = torch.nn.functional.max_pool2d(y, k, s, p, return_indices=True)
    ga = torch.randn(pooled.shape, generator=gen).to(DEV).contiguous(memory_format=cl)
    gb = torch.randn(pooled.shape, generator=gen).to(DEV).contiguous(memory_format=cl)
    ref = torch.ops.aten.max_pool2d_with_indices_backward(ga + gb, y, [k, k], [s, s], [p, p], [1, 1], False, idx)
    ref = torch.ops.aten.threshold_backward(ref, y, 0)
    idx = idx.contiguous(memory_format=cl)
    got = _hip.maxpool_bwd_relu(ga, idx, y, torch.full_like(y, float("nan")), k, s, p, gb=gb)
    again = _hip.maxpool_bwd_relu(ga, idx, y, torch.full_like(y, float("nan")), k, s, p, gb=gb)
    assert not torch.isnan(got).any() and torch.equal(got, again)
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
