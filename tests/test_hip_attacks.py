"""GPU (-m gpu): loop-level parity of the product's attack classes running on MI355X.

Two tiers:
  * identical gradient source -> bit-exact: the oracle's per-iteration gradients (CPU, reference arithmetic) are
    replayed through the HIP update kernels; every iterate and the final uint8 images must equal the
    reference's golden output (BASELINE.json configs[0]: I-FGSM / ResNet-18 / 16 images / K=10).
  * end to end (GPU surrogate forward/backward, MIOpen/rocBLAS rounding): fp32 input-gradients within 1e-5
    (relative to max|g|) of the CPU path, uint8 mismatch rate and attack-success-rate reported and bounded.
"""
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


def t(a):
    return torch.from_numpy(np.asarray(a))


def make(name, models=None, **kw):
    base = ta.load_attack_class(name)
    models = models or [backbones.create("toy_cnn", seed=3, verbose=False)]

    def load_model(self, model_name):
        wrapped = [wrap_model(m.eval().to(DEV)) for m in models]
        return wrapped[0] if len(wrapped) == 1 else EnsembleModel(wrapped)

    cls = type("Gpu" + base.__name__, (base,), {"load_model": load_model})
    atk = cls(model_name="injected", **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)      # reference's CPU draws
    return atk


def test_config1_trajectory_replay(golden):
    """configs[0] (I-FGSM / ResNet-18 / 16 images / K=10) with the oracle's gradients fed to the HIP update path:
    every iterate and the final uint8 images are bit-identical to the oracle's.  (Whether the oracle on THIS host
    also reproduces the build container's golden bytes depends on the host's oneDNN kernels; it is reported, and
    asserted transitively when it does.)"""
    g = golden("config1_ifgsm_resnet18")
    xu8 = u8_images(16, 224, int(g["seed_images"]))
    x = xu8.float() / 255
    model = backbones.create("resnet18", seed=int(g["seed_weights"]), verbose=False)
    trace = []
    delta_ref = O.run_attack("ifgsm", model, x, t(g["label"]), trace=trace)
    u8_ref = O.quantize_u8(x + delta_ref)
    host_matches_golden = np.array_equal(u8_ref, g["adv_u8"])
    print("oracle on this host reproduces the golden uint8 images: %s (mismatch %.4f%%)"
          % (host_matches_golden, 100 * float((u8_ref != g["adv_u8"]).mean())))
    xd = x.to(DEV)
    d = torch.zeros_like(xd)
    m = None
    for it, rec in enumerate(trace):
        m_out = torch.empty_like(xd)
        _hip.mi_update(rec["grad"].to(DEV), m, m_out, d, xd, 0.0, ALPHA, EPS)
        m = m_out
        assert torch.equal(d.cpu(), rec["delta"]), "iterate %d differs" % it
    out = torch.empty((16, 224, 224, 3), dtype=torch.uint8, device=DEV)
    _hip.quantize_u8_nhwc(xd, d, out)
    assert np.array_equal(out.cpu().numpy(), u8_ref)                          # final uint8 bit-exact vs oracle
    if host_matches_golden:
        assert np.array_equal(out.cpu().numpy(), g["adv_u8"])                 # ... and vs the reference's golden


def test_config1_loop_replay_default_loop(golden):
    """configs[0] through the product's I-FGSM CLASS in its default loop form -- attack.py::_forward_normalize_folded:
    ta_normalize_adv_fwd -> ResNet-18 on the device -> ta_mi_update_std -- with the oracle's gradient at the backbone's input
    injected at every iteration (``Attack.grad_inject``): the final perturbation and uint8 images are the oracle's bit for bit
    (decay = 0: sign(g / mean|g|) = sign(g), so the order of the |g| sums cannot matter), and the device's own first gradient
    agrees in sign with the reference's."""
    g = golden("config1_ifgsm_resnet18")
    x = u8_images(16, 224, int(g["seed_images"])).float() / 255
    label = t(g["label"])
    trace = []
    delta_ref = O.run_attack("ifgsm", backbones.create("resnet18", seed=int(g["seed_weights"]), verbose=False), x, label, trace=trace)
    atk = make("ifgsm", [backbones.create("resnet18", seed=int(g["seed_weights"]), verbose=False)])
    flips = []

    def inject(it, gy):
        std = atk.model[0].normalize.std.reshape(1, -1, 1, 1)
        ref = trace[it]["grad"]
        assert torch.equal(trace[it]["grads_y"][0] / std.cpu(), ref)          # Normalize's backward of the recorded gy (utils.py:76)
        flips.append(float((torch.sign((gy / std).cpu()) != torch.sign(ref)).float().mean()))
        return trace[it]["grads_y"][0].to(DEV)

    atk.grad_inject = inject
    before = dict(_hip.stats)
    delta = atk(x, label)
    assert _hip.stats["std_form_launches"] - before["std_form_launches"] == 10 == len(flips)
    print("I-FGSM / ResNet-18 / 16 images in the default loop form: device-vs-reference gradient sign flips first %.3f%% / worst "
          "%.3f%%" % (100 * flips[0], 100 * max(flips)))
    assert flips[0] <= 0.02
    assert torch.equal(delta.cpu(), delta_ref)
    u8 = quantize_images(x, delta)
    assert np.array_equal(u8, O.quantize_u8(x + delta_ref))
    if np.array_equal(O.quantize_u8(x + delta_ref), g["adv_u8"]):               # this host's oneDNN reproduces the build container's
        assert np.array_equal(u8, g["adv_u8"])                                  # -> the real reference's golden bytes


@pytest.mark.parametrize("name", ["mifgsm", "nifgsm", "tim", "sim", "admix", "dim", "dts"])
def test_trajectory_replay_reference_gradients(golden, name):
    """Reference-pinned loop parity, independent of this host's CPU: the product runs on the GPU (HIP transforms,
    surrogate forward/backward, HIP update) but each iteration's gradient is replaced by the one the REAL
    reference computed at that iteration (tests/golden/loops_toy.npz).  By induction the iterates coincide, so
      * the GPU's own fp32 input-gradient can be compared with the reference's at the same point: <= 1e-5 of max|g|;
      * the final delta must equal the reference's golden delta bit for bit."""
    g = golden("loops_toy")
    x, label = t(g["x_u8"]).float() / 255, t(g["label"])
    ref_grads = t(g["grads_" + name])
    atk = make(name)
    stats, it = [], [0]
    orig_get_grad = type(atk).get_grad

    def get_grad(self, loss, delta, **kw):
        gpu = orig_get_grad(self, loss, delta, **kw).cpu()
        ref = ref_grads[it[0]]
        diff = (gpu - ref).abs()
        stats.append((float(diff.max() / ref.abs().max()), float((gpu - ref).norm() / ref.norm()),
                      float((diff <= 1e-5 * ref.abs().max()).float().mean()),
                      float((torch.sign(gpu) != torch.sign(ref)).float().mean())))
        it[0] += 1
        return ref.to(DEV)

    type(atk).get_grad = get_grad
    torch.manual_seed(1234)
    delta = atk(x, label)
    s = np.array(stats)
    print("%s: GPU vs reference input-gradient over %d iterations: max|diff|/max|g| %.2e, rel-L2 %.2e, "
          "elements within 1e-5*max|g| %.4f%%, sign flips %.4f%%"
          % (name, len(stats), s[:, 0].max(), s[:, 1].max(), 100 * s[:, 2].min(), 100 * s[:, 3].max()))
    assert it[0] == len(ref_grads)
    # ReLU masks of a few near-zero pre-activations differ between MIOpen and oneDNN, so the max norm is not
    # 1e-5-tight; the bulk is (see test_gradient_accuracy_vs_fp64), and the signs -- all the update uses -- agree
    assert s[:, 3].max() <= 0.01 and s[:, 1].max() <= 0.1
    assert np.array_equal(delta.cpu().numpy(), g["delta_" + name])


@pytest.mark.parametrize("name", ["fgsm", "ifgsm", "mifgsm", "nifgsm", "vmifgsm", "vnifgsm", "dim", "tim", "sim",
                                  "admix", "dts", "ens"])
def test_end_to_end_gpu_vs_reference(golden, name):
    """Whole loop on the GPU (surrogate included) against the reference's golden result: invariants hold, the
    uint8 images agree up to the few pixels whose gradient sign is decided by the last bits of the surrogate's
    arithmetic, and the attack success rate on the surrogate is the same."""
    g = golden("loops_toy")
    x, label = t(g["x_u8"]).float() / 255, t(g["label"])
    models = [backbones.create("toy_cnn", seed=3, verbose=False)]
    if name == "ens":
        models.append(backbones.create("toy_cnn", seed=4, verbose=False))
    atk = make(name, models)
    torch.manual_seed(1234)
    delta = atk(x, label).cpu()
    ref = t(g["delta_" + name])
    assert float(delta.abs().max()) <= EPS + 1e-7
    adv = x + delta
    assert float(adv.min()) >= 0.0 and float(adv.max()) <= 1.0 + 1e-7
    u8_gpu = quantize_images(x, delta)
    u8_ref = O.quantize_u8(x + ref)
    mismatch = float((u8_gpu != u8_ref).mean())
    print("%s: uint8 mismatch rate GPU-vs-reference %.4f%%" % (name, 100 * mismatch))
    assert mismatch <= 0.002          # measured on MI355X: <= 0.033 % (profiles/r02/pytest_gpu_summary_r2e.txt)
    cpu_models = [m.cpu() for m in models]
    victims = cpu_models if name == "ens" else cpu_models[0]
    victim = O.logits_of(victims, t(u8_gpu).permute(0, 3, 1, 2).float() / 255)
    victim_ref = O.logits_of(victims, t(u8_ref).permute(0, 3, 1, 2).float() / 255)
    asr_gpu = float((victim.argmax(1) != label).float().mean())
    asr_ref = float((victim_ref.argmax(1) != label).float().mean())
    assert asr_gpu == asr_ref


@pytest.mark.parametrize("name,kw", [("pifgsm", {}), ("emifgsm", {}), ("iefgsm", {}), ("gnp", {}),
                                     ("gra", dict(num_neighbor=5)), ("pgn", dict(num_neighbor=4)), ("gifgsm", {}), ("dta", dict(K=3)), ("pcifgsm", {}), ("smifgrm", dict(num_neighbor=4))])
def test_more_gradient_attacks_gpu_vs_reference(golden, name, kw):
    """SURVEY 8(f) rank 3 on the GPU: PI / EMI / IE-FGSM, GNP, GRA, PGN end to end against the reference's golden."""
    g, base = golden("loops_more"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    atk = make(name, **kw)
    torch.manual_seed(1234)
    delta = atk(x, label).cpu()
    assert float(delta.abs().max()) <= EPS + 1e-7
    mismatch = float((quantize_images(x, delta) != O.quantize_u8(x + t(g["delta_" + name]))).mean())
    print("%s: uint8 mismatch rate GPU-vs-reference %.4f%%" % (name, 100 * mismatch))
    assert mismatch <= 0.002          # measured on MI355X: <= 0.033 % (profiles/r02/pytest_gpu_summary_r2e.txt)


@pytest.mark.parametrize("name,kw", [("mig", dict(s_factor=5)), ("aifgtm", {}), ("mef", dict(num_neighbor=4, epoch=6)),
                                     ("gaa", dict(N=3, epoch=5)), ("dem", {})])
def test_long_tail_attacks_gpu_vs_reference(golden, name, kw):
    """MIG / AI-FGTM / MEF / GAA / DEM end to end on the GPU against the reference's golden loops"""
    g, base = golden("loops_tail"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    atk = make(name, **kw)
    torch.manual_seed(1234)
    delta = atk(x, label).cpu()
    assert float(delta.abs().max()) <= EPS + 1e-7
    mismatch = float((quantize_images(x, delta) != O.quantize_u8(x + t(g["delta_" + name]))).mean())
    print("%s: uint8 mismatch rate GPU-vs-reference %.4f%%" % (name, 100 * mismatch))
    assert mismatch <= 0.005


def test_variants_run_on_gpu(golden):
    g = golden("loops_toy")
    x, label = t(g["x_u8"]).float() / 255, t(g["label"])
    import os
    before = dict(_hip.stats)
    x224 = u8_images(2, 224, 9).float() / 255                             # 224 px: no resize in PreprocessingModel
    make("mifgsm")(x224, label[:2])                                       # -> the loop with the Normalize folded into its ends:
    assert _hip.stats["std_form_launches"] - before["std_form_launches"] == 10   # ta_mi_update_std; this backbone's backward is
    assert _hip.stats["k1_passes"] - before["k1_passes"] == 10            # not ours, so a sum-only pass over gy precedes it
    old = os.environ.get("TA_FOLD_NORMALIZE")
    os.environ["TA_FOLD_NORMALIZE"] = "0"                                 # the hook-by-hook loop: the Normalize backward is the
    try:                                                                  # producer of g and feeds all 10 fused updates
        mid = dict(_hip.stats)
        make("mifgsm")(x224, label[:2])
        assert _hip.stats["partials_reused"] - mid["partials_reused"] == 10 and _hip.stats["k1_passes"] == mid["k1_passes"]
    finally:
        os.environ.pop("TA_FOLD_NORMALIZE") if old is None else os.environ.__setitem__("TA_FOLD_NORMALIZE", old)
    before = dict(_hip.stats)
    make("mifgsm")(x, label)                                              # 32 px -> resized to 224: bilinear backward is
    assert _hip.stats["k1_passes"] - before["k1_passes"] == 10            # the producer, the update runs its own K1
    d = make("mifgsm", targeted=True)(x, [label, t(g["target"])])
    assert float(d.abs().max()) <= EPS + 1e-7
    atk = make("mifgsm", random_start=True)
    atk.noise_source = None                                               # in-kernel Philox random start
    d1 = atk(x, label)
    atk2 = make("mifgsm", random_start=True)
    atk2.noise_source = None
    assert torch.equal(d1, atk2(x, label))                                # same seed, same offset -> same stream
    atk = make("vmifgsm", num_neighbor=3, epoch=2)
    atk.noise_source = None
    assert float(atk(x, label).abs().max()) <= EPS + 1e-7
    atk = make("mifgsm", norm="l2", epsilon=3.0, alpha=0.3)
    d = atk(x, label)
    assert float(d.flatten(1).norm(dim=1).max()) <= 3.0 * (1 + 1e-5)


def _loop(monkeypatch, fold, name, backbone=None, model_name=None, n=3, **kw):
    """one attack loop on seeded 224-pixel images with the surrogate's Normalize folded into the loop's two ends or not"""
    monkeypatch.setenv("TA_FOLD_NORMALIZE", "1" if fold else "0")
    x = u8_images(n, 224, 77).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(78))
    if model_name is not None:                                   # through Attack.load_model (the bench's arrangement)
        atk = ta.load_attack_class(name)(model_name=model_name, **kw)
    else:
        atk = make(name, [backbones.create(backbone, seed=5, verbose=False)], **kw)
    torch.manual_seed(99)                                        # the random start's CPU draws (noise_source)
    stats = dict(_hip.stats)
    delta = atk(x, label).cpu()
    return delta, {k: _hip.stats[k] - stats[k] for k in stats}


@pytest.mark.parametrize("name,backbone,kw", [("mifgsm", "toy_cnn", dict(epoch=4)), ("ifgsm", "toy_cnn", dict(epoch=3)),
                                               ("fgsm", "toy_cnn", {}), ("mifgsm", "toy_cnn", dict(epoch=3, random_start=True)),
                                               ("mifgsm", "vit_tiny_patch16_224", dict(epoch=3))])
def test_normalize_folded_loop_equals_hook_loop(monkeypatch, name, backbone, kw):
    """attack.py::_forward_normalize_folded (ta_normalize_adv_fwd -> backbone -> ta_mi_update_std) against the hook-by-hook
    loop (add, Normalize, backbone, Normalize backward with its |g| sums, ta_mi_update with x_adv): the same rounding points,
    and with the sums taken by K1 in ta_normalize_bwd's order the same bits -- the final delta is EQUAL wherever the surrogate
    itself is run-to-run deterministic (measured by running the hook loop twice; else the folded loop may differ from it no
    more than it differs from itself)."""
    plain, st0 = _loop(monkeypatch, False, name, backbone, **kw)
    again, _ = _loop(monkeypatch, False, name, backbone, **kw)
    folded, st1 = _loop(monkeypatch, True, name, backbone, **kw)
    folded_again, _ = _loop(monkeypatch, True, name, backbone, **kw)
    k = kw.get("epoch", 1 if name == "fgsm" else 10)
    assert st0["std_form_launches"] == 0 and st1["std_form_launches"] == k
    assert float(folded.abs().max()) > 0 and float(folded.abs().max()) <= EPS + 1e-7
    noise = max(float((plain != again).float().mean()), float((folded != folded_again).float().mean()))
    diff = float((plain != folded).float().mean())
    print("%s / %s: folded vs hook loop differ in %.5f%% of the elements (hook loop vs itself: %.5f%%)" % (name, backbone, 100 * diff, 100 * noise))
    assert diff <= 3 * noise + (0.0 if noise == 0.0 else 1e-4)


def _recorded_loop(monkeypatch, fold, model_name, n, epoch):
    """MI-FGSM through ``Attack.load_model`` (bench.py's arrangement comes from the environment) with every iteration's
    (delta, momentum) after the fused update recorded -> (list of (delta, momentum) on the CPU, launch statistics)"""
    monkeypatch.setenv("TA_FOLD_NORMALIZE", "1" if fold else "0")
    x = u8_images(n, 224, 77).float() / 255
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(78))
    atk = ta.load_attack_class("mifgsm")(model_name=model_name, epoch=epoch)
    log, inner = [], atk._fused_update

    def recording(grad, momentum, delta, data, **kw):
        m = inner(grad, momentum, delta, data, **kw)
        log.append((delta.detach().cpu().clone(), m.detach().cpu().clone()))
        return m

    atk._fused_update = recording
    stats = dict(_hip.stats)
    atk(x, label)
    return log, {k: _hip.stats[k] - stats[k] for k in stats}


@pytest.mark.parametrize("model_name", ["resnet18", "resnet50"])
def test_normalize_folded_loop_fused_resnet(monkeypatch, model_name):
    """The loop bench.py times (folded BatchNorm, NHWC, fused glue, stem kernel leaving the sums of |gy / std|, ta_mi_update_std)
    against the hook-by-hook loop of the same arrangement (Normalize module, ta_normalize_bwd with its |g| sums, ta_mi_update),
    K = 10, under ``TA_DETERMINISTIC=1`` -- the surrogate is then run-to-run deterministic (asserted: the hook loop reproduces
    itself bit for bit), so the comparison has NO noise term.  Both loops see the same gy; the stem kernel adds its sums in
    another order than ta_normalize_bwd, so g / mean|g| and the momentum may differ in the last bit and a momentum within
    rounding of zero may take the other sign.  The rule is test_hip_kernels.py::assert_delta_equal's, applied iterate by
    iterate: every iterate EQUAL bit for bit up to the first one that differs, and that one differs only where the hook loop's
    momentum is within 1e-5 of zero, in <= 1e-6 of the elements (from there on a chaotic random-init network legitimately
    follows another trajectory; what remains is printed).  Measured: no iterate differs at all."""
    from test_hip_kernels import assert_delta_equal
    for k_, v in (("TA_FOLD_BN", "1"), ("TA_CHANNELS_LAST", "1"), ("TA_FUSED_GLUE", "1"), ("TA_STEM_KERNEL", "1"),
                  ("TA_ALLOW_RANDOM_INIT", "1"), ("TA_DETERMINISTIC", "1")):
        monkeypatch.setenv(k_, v)
    try:
        plain, st0 = _recorded_loop(monkeypatch, False, model_name, 2, 10)
        folded, st1 = _recorded_loop(monkeypatch, True, model_name, 2, 10)
        again, _ = _recorded_loop(monkeypatch, False, model_name, 2, 10) if model_name == "resnet18" else (plain, None)
    finally:
        monkeypatch.setenv("TA_DETERMINISTIC", "0")
        ta.attack.deterministic_mode()                                   # hands the process-wide torch flags back
    assert st1["std_form_launches"] == 10 and st1["partials_reused"] == 10 and st1["k1_passes"] == 0
    assert st0["std_form_launches"] == 0 and st0["k1_passes"] == 0 and st0["partials_reused"] == 10
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(plain, again)), \
        "TA_DETERMINISTIC=1: the hook loop does not reproduce itself"
    first = next((k for k in range(10) if not torch.equal(plain[k][0], folded[k][0])), None)
    for k in range(10 if first is None else first):          # same iterates in -> momenta equal to the sums' rounding: each of
        err = float((folded[k][1] - plain[k][1]).abs().max())           # the k + 1 terms g / mean|g| within a few ulp
        assert err <= 1e-6 * (k + 1) * float(plain[k][1].abs().max()), "iteration %d: momenta differ by %.2e" % (k, err)
    if first is not None:
        assert_delta_equal(folded[first][0].numpy(), plain[first][0].numpy(), plain[first][1].numpy())
    final = float((plain[-1][0] != folded[-1][0]).float().mean())
    print("fused %s, K=10, deterministic: first iterate at which the folded loop's delta differs from the hook loop's: %s; final "
          "delta differs in %.5f%% of the elements" % (model_name, first, 100 * final))
    assert float(folded[-1][0].abs().max()) > 0 and float(folded[-1][0].abs().max()) <= EPS + 1e-7


@pytest.mark.parametrize("backbone", ["toy_cnn"])          # (vit_tiny_patch16_224 ran equal too, r5b / r5k: 42 s of the tier's time)
def test_vmi_neighbours_stacked_equal_one_by_one(monkeypatch, backbone):
    """VMI-FGSM's folded loop with k neighbour samples per surrogate evaluation (gradient/vmifgsm.py::_neighbor_stack)
    against one evaluation per neighbour (vmifgsm.py:46-58's shape): same Philox draws, same accumulation order, per-slice
    batch-mean losses -- the final delta may differ only where the surrogate's libraries round a k * N batch differently."""
    out = {}
    for k in ("1", "4", "2"):
        monkeypatch.setenv("TA_VMI_STACK", k)
        torch.manual_seed(5)
        before = _hip.stats["partials_reused"]
        out[k], _ = _loop(monkeypatch, True, "vmifgsm", backbone, n=2, epoch=2, num_neighbor=4)
        assert _hip.stats["partials_reused"] == before + 2
    for k in ("4", "2"):
        diff = float((out[k] != out["1"]).float().mean())
        print("VMI-FGSM / %s: %s neighbours per evaluation vs one: delta differs in %.5f%% of the elements" % (backbone, k, 100 * diff))
        assert diff <= 2e-4
    assert float(out["4"].abs().max()) > 0


def conditioned_fixture(golden):
    """(images, labels, fixture, mask of comparable elements, builder of the conditioned ResNet-50): the surrogate of
    oracle/gen_conditioned.py -- the seeded ResNet-50 with gamma * 0.2 on every block's last BatchNorm and every ReLU-feeding
    BatchNorm bias moved so that NO pre-activation of the two fixture images is within 2.7e-4 (relative) of zero: the input
    gradient is then a smooth function of the arithmetic (the reference's fp32 CPU path is 9.7e-7 from fp64 on it).  The mask
    excludes the 11 x 11 input patches under the listed max-pool windows whose two best candidates are within 2e-5."""
    import gen_conditioned as GC
    g = golden("conditioned_resnet50")
    x = u8_images(GC.N, 224, int(g["seed_images"])).float() / 255
    mask = torch.ones(x.shape, dtype=torch.bool)
    for n_, _c, ph, pw in g["ties"].tolist():
        mask[n_, :, max(0, 4 * ph - 5):4 * ph + 6, max(0, 4 * pw - 5):4 * pw + 6] = False
    return x, t(g["label"]), g, mask, (lambda: GC.conditioned_resnet50(g["bias_moves"]))


def conditioned_gradient(build, x, label, fold_bn, channels_last, fold_normalize, monkeypatch):
    """the product's own first-iteration input gradient (attack.py:118-122) on DEV in the given arrangement"""
    from transferattack_amd.attack import takes_channels_last
    monkeypatch.setenv("TA_FOLD_NORMALIZE", "1" if fold_normalize else "0")
    base = ta.load_attack_class("mifgsm")

    def load_model(self, model_name):
        net = build()
        for p in net.parameters():
            p.requires_grad_(False)
        if fold_bn:
            backbones.fold_batchnorm(net)
        w = wrap_model(net.eval().to(DEV))
        return w.to(memory_format=torch.channels_last) if channels_last and takes_channels_last(net) else w

    atk = type("Cond" + base.__name__, (base,), {"load_model": load_model})(model_name="injected", epoch=1)
    got = []
    if fold_normalize:
        atk.grad_probe = lambda it, grad: got.append(grad.detach().clone())
    else:
        inner = base.get_grad
        type(atk).get_grad = lambda self, loss, delta, **kw: (got.append(inner(self, loss, delta, **kw)), got[-1])[1]
    before = _hip.stats["std_form_launches"]
    atk(x, label)
    assert (_hip.stats["std_form_launches"] == before + 1) == bool(fold_normalize)
    return got[0].cpu()


CONDITIONED_ARRANGEMENTS = [("reference-literal (NCHW, separate BatchNorm, hook loop)", False, False, False),
                            ("reference-literal, Normalize folded into the loop's ends", False, False, True),
                            ("bench arrangement (folded BatchNorm, NHWC, fused glue, stem kernel, folded Normalize)", True, True, True),
                            ("bench arrangement with the glue in the convolutions' epilogues (CK)", True, True, True)]


@pytest.mark.parametrize("tag,fold_bn,channels_last,fold_normalize", CONDITIONED_ARRANGEMENTS)
def test_gradient_within_1e5_on_conditioned_resnet50(golden, monkeypatch, tag, fold_bn, channels_last, fold_normalize):
    """SURVEY 8 row a5 / north_star "fp32 grads within 1e-5", on the configs[1] surrogate: every comparable element of the
    product's input gradient is within 1e-5 * max|g| of the REAL reference's CPU gradient (oracle/gen_conditioned.py recorded
    its ``Attack.get_grad``), in the reference-literal and in bench.py's arrangement.  The plain seeded ResNet-50 cannot carry
    this bound on ANY pair of fp32 implementations (test_gradient_accuracy_vs_fp64 prints its numbers): this fixture removes
    the ReLU / max-pool discontinuities and the amplification that hide an implementation's own error behind them."""
    x, label, g, mask, build = conditioned_fixture(golden)
    monkeypatch.setenv("TA_CK_EPILOGUE", "1" if "(CK)" in tag else "0")
    g_ref = t(g["grad_reference_cpu_fp32"])
    scale = float(g_ref.abs().max())
    got = conditioned_gradient(build, x, label, fold_bn, channels_last, fold_normalize, monkeypatch)
    diff = (got.double() - g_ref.double()).abs() / scale
    rel = float((got.double() - g_ref.double()).norm() / g_ref.double().norm())
    inside = float((diff <= 1e-5).float().mean())
    print("conditioned ResNet-50, %s: max |g - g_ref| / max|g_ref| = %.2e over the comparable elements (%.2e over all; %.4f%% "
          "of ALL elements within 1e-5), rel-L2 %.2e; the reference's own fp32 path is %.2e / %.2e from fp64"
          % (tag, float(diff[mask].max()), float(diff.max()), 100 * inside, rel, float(g["max_reference_vs_fp64"]),
             float(g["rel_l2_reference_vs_fp64"])))
    assert float(mask.float().mean()) >= 0.99
    assert float(diff[mask].max()) <= 1e-5, "the input gradient leaves the 1e-5 band of north_star"
    assert float((torch.sign(got) != torch.sign(g_ref))[mask].float().mean()) <= 1e-4


@pytest.mark.parametrize("name,n", [("toy_cnn", 4), ("resnet18", 4), ("resnet50", 4)])
def test_gradient_accuracy_vs_fp64(name, n):
    """How far is the MI355X fp32 input-gradient (MIOpen / rocBLAS) from the exact gradient, compared with how far
    the reference's own fp32 CPU path (oneDNN) is?  Truth = the same network in fp64 on the CPU.  The GPU path must
    be as accurate as the reference's (<= 4x its relative L2 error, or 1e-5), and agree with it on the sign of
    >= 99% of the elements; the distribution is printed for DESIGN.md."""
    size = 32 if name == "toy_cnn" else 224
    classes = 10 if name == "toy_cnn" else 1000
    x = u8_images(n, size, 5).float() / 255
    label = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(6))
    model = backbones.create(name, seed=0, verbose=False)

    def grad_cpu(dtype):
        m = backbones.create(name, seed=0, verbose=False).to(dtype)
        d = torch.zeros_like(x, dtype=dtype, requires_grad=True)
        cfg = O.preprocess_cfg(m)
        logits = m(O.preprocess(x.to(dtype) + d, cfg[0], [float(v) for v in cfg[1]], [float(v) for v in cfg[2]]).to(dtype))
        return torch.autograd.grad(torch.nn.functional.cross_entropy(logits, label), d)[0]

    g64 = grad_cpu(torch.float64)
    g32 = grad_cpu(torch.float32)
    atk = make("mifgsm", [model])
    grads = []
    orig = type(atk).get_grad

    def get_grad(self, loss, delta, **kw):
        grads.append(orig(self, loss, delta, **kw))
        return grads[-1]

    type(atk).get_grad = get_grad
    atk.epoch = 1
    atk(x, label)
    ggpu = grads[0].cpu()
    err = lambda a: float((a.double() - g64).norm() / g64.norm())      # noqa: E731
    e_cpu, e_gpu = err(g32), err(ggpu)
    within = float(((ggpu - g32).abs() <= 1e-5 * g32.abs().max()).float().mean())
    flips = float((torch.sign(ggpu) != torch.sign(g32)).float().mean())
    print("%s: rel-L2 error vs fp64 truth: CPU fp32 %.3e, MI355X fp32 %.3e; GPU-vs-CPU elements within 1e-5*max|g| "
          "%.4f%%, sign flips %.4f%%" % (name, e_cpu, e_gpu, 100 * within, 100 * flips))
    # the 224-pixel random-init networks get the absolute floor of test_fold_bn_channels_last_is_the_same_surrogate: the
    # device's error is a run-to-run noisy quantity there, and a ratio of two such numbers must not carry a -x tier
    # floors per surrogate (round 5, ADVICE): 1.5x the worst device figure of rounds 2-4 instead of one 3e-2 for all
    assert e_gpu <= max(4 * e_cpu, {"toy_cnn": 1e-5, "resnet18": 1e-2, "resnet50": 2.5e-2}[name])
    assert flips <= 0.01


def test_main_cli_roundtrip(tmp_path, monkeypatch):
    """main.py with the reference's flags: PNGs written = floor((x + delta) * 255) of the attack's output, and the
    --eval pass reads them back (ASR row format of main.py:72-77)."""
    import csv
    import sys
    from PIL import Image
    import main as cli
    inp, out = tmp_path / "data", tmp_path / "adv"
    (inp / "images").mkdir(parents=True)
    xu8 = u8_images(5, 224, 11).permute(0, 2, 3, 1).numpy()
    with open(inp / "labels.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["filename", "label", "targeted_label"])
        for i in range(5):
            Image.fromarray(xu8[i]).save(inp / "images" / ("%d.png" % i))
            w.writerow(["%d.png" % i, i, i + 1])
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, "argv", ["main.py", "--input_dir", str(inp), "--output_dir", str(out), "--attack", "mifgsm",
                                      "--model", "toy_cnn", "--batchsize", "4"])
    cli.main()
    x = torch.from_numpy(xu8).permute(0, 3, 1, 2).float() / 255
    atk = ta.load_attack_class("mifgsm")(model_name="toy_cnn")
    for lo, hi in ((0, 4), (4, 5)):
        delta = atk(x[lo:hi], torch.arange(lo, hi))
        want = quantize_images(x[lo:hi], delta)
        for i in range(lo, hi):
            assert np.array_equal(np.array(Image.open(out / ("%d.png" % i))), want[i - lo])
    monkeypatch.setattr(cli, "cnn_model_paper", ["resnet18"])
    monkeypatch.setattr(cli, "vit_model_paper", [])
    monkeypatch.setattr(sys, "argv", ["main.py", "--input_dir", str(inp), "--output_dir", str(out), "--eval"])
    monkeypatch.delenv("TA_WEIGHTS_DIR", raising=False)
    with pytest.raises(SystemExit, match="pretrained weights"):          # no silent ASR against random-init victims
        cli.main()
    monkeypatch.setenv("TA_ALLOW_RANDOM_INIT", "1")                      # plumbing test: explicitly allowed
    cli.main()
    assert "|" in open(tmp_path / "results_eval.txt").read()
    monkeypatch.setenv("TA_WEIGHTS_DIR", str(tmp_path))                  # weights directory without the file: refuse
    monkeypatch.delenv("TA_ALLOW_RANDOM_INIT")
    with pytest.raises(FileNotFoundError):
        backbones.create("resnet18", verbose=False)


def test_deterministic_mode_writes_identical_pngs(tmp_path):
    """TA_DETERMINISTIC=1 (attack.py::deterministic_mode): two separate processes running configs[1]'s attack (MI-FGSM,
    ResNet-50, K=10) on one 32-image reference batch in bench.py's arrangement write byte-identical PNGs -- the reference's
    contract is "final uint8 bit-exact" (utils.py:63-66), and without the switch MIOpen's atomics make the same command differ
    from itself in ~0.3 % of the gradient signs (test_fused_surrogate_path_on_device prints the figure).  Also printed: what
    the switch costs in images/s at this batch."""
    import csv
    import subprocess
    import sys
    import time
    from PIL import Image
    inp = tmp_path / "data"
    (inp / "images").mkdir(parents=True)
    xu8 = u8_images(32, 224, 12).permute(0, 2, 3, 1).numpy()
    with open(inp / "labels.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["filename", "label", "targeted_label"])
        for i in range(32):
            Image.fromarray(xu8[i]).save(inp / "images" / ("%02d.png" % i))
            w.writerow(["%02d.png" % i, (37 * i) % 1000, (37 * i + 1) % 1000])
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ, TA_FOLD_BN="1", TA_CHANNELS_LAST="1", TA_ALLOW_RANDOM_INIT="1", PYTHONPATH=root)

    def run(tag, deterministic):
        out = tmp_path / tag
        env = dict(base, TA_DETERMINISTIC="1" if deterministic else "0")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(root, "main.py"), "--input_dir", str(inp), "--output_dir", str(out),
                            "--attack", "mifgsm", "--model", "resnet50", "--batchsize", "32", "--profile"], env=env,
                           cwd=str(tmp_path), capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        return {f: open(out / f, "rb").read() for f in sorted(os.listdir(out))}, time.perf_counter() - t0, r.stdout

    first, t1, _ = run("det1", True)
    second, t2, log2 = run("det2", True)
    assert len(first) == 32 and first.keys() == second.keys()
    differing = [f for f in first if first[f] != second[f]]
    loose1, _, _ = run("plain1", False)
    loose2, t4, log4 = run("plain2", False)
    px = lambda blobs, f: np.array(Image.open(__import__("io").BytesIO(blobs[f])))       # noqa: E731
    noise = float(np.mean([np.mean(px(loose1, f) != px(loose2, f)) for f in loose1]))
    moved = float(np.mean([np.mean(px(first, f) != px(loose1, f)) for f in first]))
    grab = lambda log: [ln for ln in log.splitlines() if ln.startswith("{")][-1][:300] if "{" in log else ""   # noqa: E731
    print("TA_DETERMINISTIC=1: %d of 32 PNGs differ between two processes (asserted 0); without it two processes differ in "
          "%.4f%% of the uint8 values; deterministic vs default algorithms: %.4f%% of the values; process wall %.1f s (first, "
          "incl. MIOpen kernel builds) / %.1f s deterministic, %.1f s default\n  deterministic: %s\n  default: %s"
          % (len(differing), 100 * noise, 100 * moved, t1, t2, t4, grab(log2), grab(log4)))
    assert not differing, differing[:5]
    assert any(first[f] != open(inp / "images" / f, "rb").read() for f in first)       # it did attack


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


@pytest.mark.parametrize("name", ["svre", "cwa"])
def test_per_member_ensemble_attacks_gpu(golden, name):
    """SURVEY 8(f) rank 4 on the GPU: SVRE / CWA over EnsembleModel.models[k] vs the reference's golden result."""
    g, base = golden("loops_more"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
    cls = ta.load_attack_class(name)

    def load_model(self, model_name):
        return EnsembleModel([wrap_model(m.eval().to(DEV)) for m in models])

    atk = type("Gpu" + cls.__name__, (cls,), {"load_model": load_model})(model_name=["a", "b"])
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    torch.manual_seed(1234)
    np.random.seed(99)
    delta = atk(x, label).cpu()
    assert float(delta.abs().max()) <= EPS + 1e-7
    mismatch = float((quantize_images(x, delta) != O.quantize_u8(x + t(g["delta_" + name]))).mean())
    print("%s: uint8 mismatch rate GPU-vs-reference %.4f%%" % (name, 100 * mismatch))
    assert mismatch <= 0.002          # measured on MI355X: <= 0.033 % (profiles/r02/pytest_gpu_summary_r2e.txt)


def test_ensemble_members_on_streams(monkeypatch):
    """EnsembleModel runs member k on HIP stream k (utils.py: _members_on_streams; autograd runs each member's backward on
    that stream too): same kernels, same per-member order -> the logits and the summed input gradient of the one-stream run,
    repeated to give an ordering bug between the streams a chance to show.  Two ResNet-18 + ViT-B/16 at 224 px (the
    stem / glue / Normalize kernels and the |g| sums of ta_sum_members included); the bound is 10x the run-to-run spread of
    the one-stream run itself (MIOpen's atomically accumulated backward-data kernels), at least 1e-3 of the gradient's norm."""
    # (two ResNet-18 with different weights + the ViT: three members whose kernels MIOpen has mostly met earlier in the tier --
    # r4e's version with VGG-16 spent 59 s, most of it in MIOpen's find for VGG's convolutions at batch 4)
    names = (("resnet18", 0), ("resnet18", 1), ("vit_base_patch16_224", 0))
    nets = [wrap_model(backbones.create(n, seed=sd, verbose=False).eval().to(DEV)) for n, sd in names]
    for net in nets:
        for p in net.parameters():
            p.requires_grad_(False)
    ens = EnsembleModel(nets)
    x = (u8_images(4, 224, 9).float() / 255).to(DEV)
    label = torch.randint(0, 1000, (4,), generator=torch.Generator().manual_seed(2)).to(DEV)

    def run():
        xin = x.clone().requires_grad_(True)
        logits = ens(xin)
        grad = torch.autograd.grad(torch.nn.functional.cross_entropy(logits, label), xin)[0]
        sums = _hip.partials_of(grad)
        assert sums is not None, "ta_sum_members left no |g| sums"
        total = sums[0][:4 * sums[1]].view(4, -1).sum(1)
        return logits.detach().clone(), grad.clone(), total.clone()

    monkeypatch.setenv("TA_ENS_STREAMS", "0")
    logits0, grad0, sums0 = run()
    spread = max(float((run()[1] - grad0).norm() / grad0.norm()) for _ in range(3))      # the one-stream run against itself
    assert torch.allclose(sums0.double(), grad0.double().abs().flatten(1).sum(1), rtol=1e-4)
    monkeypatch.setenv("TA_ENS_STREAMS", "1")
    worst = 0.0
    for _ in range(6):
        logits1, grad1, sums1 = run()
        assert float((logits1 - logits0).abs().max()) <= 1e-5 * float(logits0.abs().max())
        worst = max(worst, float((grad1 - grad0).norm() / grad0.norm()))
        assert torch.allclose(sums1.double(), grad1.double().abs().flatten(1).sum(1), rtol=1e-4)
    torch.cuda.synchronize()
    print("ensemble on 3 member streams vs one stream: input-gradient rel-L2 difference <= %.2e over 6 runs (one stream "
          "against itself: %.2e)" % (worst, spread))
    assert worst <= max(1e-3, 10 * spread)             # an ordering bug between the streams reads garbage: O(1), not O(1e-3)
