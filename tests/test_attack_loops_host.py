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
import test_hip_loops_golden as L
from transferattack_amd import _hip


@pytest.fixture(autouse=True)
def host_backend(monkeypatch):
    host_kernels.install(monkeypatch)
    monkeypatch.setattr(A, "DEV", "cpu")
    monkeypatch.setattr(L, "DEV", "cpu")


import os

FULL = os.environ.get("TA_HOST_FULL", "0") == "1"        # 1 = every case of the GPU tier in the default sum order too (+40 s)


def _subset(cases, keep):
    return cases if FULL else [c for c in cases if (c[0] if isinstance(c, tuple) else c) in keep]


test_config1_trajectory_replay = A.test_config1_trajectory_replay
test_config1_loop_replay_default_loop = A.test_config1_loop_replay_default_loop


@pytest.mark.parametrize("name,backbone,kw", [("mifgsm", "toy_cnn", dict(epoch=4)), ("ifgsm", "toy_cnn", dict(epoch=3)),
                                               ("mifgsm", "toy_cnn", dict(epoch=3, random_start=True))])
def test_normalize_folded_loop_equals_hook_loop(monkeypatch, name, backbone, kw):
    A.test_normalize_folded_loop_equals_hook_loop(monkeypatch, name, backbone, kw)


def test_normalize_folded_loop_fused_resnet(monkeypatch):
    A.test_normalize_folded_loop_fused_resnet(monkeypatch, "resnet18")


def test_vmi_neighbours_stacked_equal_one_by_one(monkeypatch):
    A.test_vmi_neighbours_stacked_equal_one_by_one(monkeypatch, "toy_cnn")


@pytest.mark.parametrize("tag,fold_bn,channels_last,fold_normalize", A.CONDITIONED_ARRANGEMENTS[:3])
def test_gradient_within_1e5_on_conditioned_resnet50(golden, monkeypatch, tag, fold_bn, channels_last, fold_normalize):
    """the GPU tier's a5 test through the kernels' host build (ATen's CPU convolutions on both sides)"""
    A.test_gradient_within_1e5_on_conditioned_resnet50(golden, monkeypatch, tag, fold_bn, channels_last, fold_normalize)
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


@pytest.mark.parametrize("name,kw", [("mig", dict(s_factor=5)), ("aifgtm", {}), ("mef", dict(num_neighbor=4, epoch=6)),
                                     ("gaa", dict(N=3, epoch=5)), ("dem", {})])
def test_long_tail_attacks_vs_reference(golden, name, kw):
    A.test_long_tail_attacks_gpu_vs_reference(golden, name, kw)


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
    monkeypatch.setattr(W, "BOUND", 0.001)                # the spectrum view is the MFMA product form: the reference's transform
    W.test_fgsra(golden)                                  # to fp32 rounding, so a few momentum signs near zero may differ


def test_sia_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)
    W.test_sia_attack(golden)


def test_ssm_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.003)                # as for FGSRA: product form of the DCT pair (measured 0.11 %)
    W.test_ssm_attack(golden)


def test_bsr_loop(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0)
    W.test_bsr_attack(golden)


@pytest.mark.parametrize("name,kw", [("ifgssm", {}), ("vaifgsm", dict(epoch=4)), ("adamsi_fgm", {}),
                                     ("rgmifgsm", dict(num_directions=2, pre_epoch=2, epoch=4)), ("dual_mifgsm", dict(epoch=5)),
                                     ("ens_mifgsm", dict(epoch=3, num_d=2)), ("maskblock", dict(patch_size=16)),
                                     ("usmm", dict(num_scale=3, num_mix=2)), ("anda", dict(n_ens=4, epoch=3)),
                                     ("rap", dict(epoch=6, transpoint=3, adv_steps=2)), ("decowa", dict(num_warping=3, epoch=3)),
                                     ("foolmix", dict(epoch=4, m=3, n=2, k=3, grad_chunk_size=5, print_timing=False)),
                                     ("ops", dict(num_sample_neighbor=2, num_sample_operator=3, epoch=2))])
def test_more_attacks_through_kernels(golden, monkeypatch, name, kw):
    """the not-yet-measured GPU test of these attacks, run through the kernels' own code on the host"""
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.0025)               # half the GPU bounds: only the kernels' fixed-order sum|g| differs
    if name == "ops":
        W._run_more_attack(golden, name, kw, W.BOUND)
    else:
        W.test_more_attacks_gpu_vs_reference(golden, name, kw)


@pytest.mark.parametrize("name,kw", [("ssm_h", dict(num_spectrum=2, epoch=2)), ("ssm_p", dict(num_scale=4, epoch=3))])
def test_ssm_tricks_through_kernels(golden, monkeypatch, name, kw):
    """... and of SSM_H / SSM_P: every view through the MFMA kernel's code (forward and backward) on the host"""
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.002)
    W.test_ssm_tricks_gpu_vs_reference(golden, name, kw)


def test_su_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.002)
    W.test_su_gpu_vs_reference(golden)


def test_everywhere_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.002)
    W.test_everywhere_gpu_vs_reference(golden)


def test_l2t_through_kernels(golden, monkeypatch):
    import test_zz_hip_widened as W
    monkeypatch.setattr(W, "DEV", "cpu")
    monkeypatch.setattr(W, "BOUND", 0.002)
    W.test_l2t_gpu_vs_reference(golden)


def test_main_cli_roundtrip(tmp_path, monkeypatch):
    """main.py end to end (decode -> attack -> quantise -> PNG -> --eval) with the kernels' own code on the host"""
    A.test_main_cli_roundtrip(tmp_path, monkeypatch)


def test_main_cli_resume(tmp_path, monkeypatch):
    A.test_main_cli_resume(tmp_path, monkeypatch)


def test_config1_end_to_end_through_kernels(golden):
    """BASELINE.json configs[0] -- I-FGSM, ResNet-18, 16 images, eps = 16/255, K = 10 -- END TO END (no replayed
    gradients): the product's attack class, the binding and the kernel sources on the host, the surrogate on torch's CPU
    path.  With the surrogate's arithmetic equal to the reference's the written uint8 images must be the reference's
    golden bytes exactly (for I-FGSM the step depends on sign(g) only, so not even the sum|g| order can matter)."""
    from conftest import u8_images
    from transferattack_amd import backbones
    from transferattack_amd.utils import quantize_images, wrap_model
    import transferattack_amd as ta
    g = golden("config1_ifgsm_resnet18")
    x = u8_images(16, 224, int(g["seed_images"])).float() / 255
    model = backbones.create("resnet18", seed=int(g["seed_weights"]), verbose=False)
    cls = ta.load_attack_class("ifgsm")
    atk = type("HostIFGSM", (cls,), {"load_model": lambda self, mn: wrap_model(model.eval())})(model_name="injected")
    delta = atk(x, A.t(g["label"]))
    out = quantize_images(x, delta)
    rate = float((out != g["adv_u8"]).mean())
    print("configs[0] end to end through the kernel sources: uint8 mismatch vs the reference's golden images %.4f%%"
          % (100 * rate))
    assert rate == 0.0


def test_config2_end_to_end_through_kernels(golden):
    """BASELINE.json configs[1] in miniature -- MI-FGSM, ResNet-50 (seeded init), K = 10, four of the synthetic images --
    end to end through the kernel sources with the surrogate on torch's CPU path, against the images the reference's
    own class wrote.  Unlike configs[0] the momentum matters here: the update kernels add |g| in their own fixed order,
    ATen in its AVX2 cascade order (itself a property of the CPU the reference happens to run on), so g / mean|g| can
    differ in the last bit.  What is asserted: the first iterate agrees with the reference everywhere except where that
    last bit decides a sign (a handful of pixels at most), and the result obeys the attack's invariants.  What is
    printed: the final mismatch -- the seeded-random ResNet-50 amplifies a handful of flipped pixels over ten
    iterations exactly as it amplifies MIOpen-vs-oneDNN rounding on the device; with the oracle-backed
    binding (ATen's order) the same loop reproduces the golden bytes exactly (tests/test_host_logic.py)."""
    from conftest import u8_images
    from transferattack_amd import backbones
    from transferattack_amd.utils import quantize_images, wrap_model
    import transferattack_amd as ta
    g = golden("config2_mifgsm_resnet50_n4")
    x = u8_images(4, 224, int(g["seed_images"])).float() / 255
    label = A.t(g["label"])
    model = backbones.create("resnet50", seed=int(g["seed_weights"]), verbose=False)
    cls = ta.load_attack_class("mifgsm")
    make_attack = lambda **kw: type("HostMIFGSM", (cls,), {"load_model": lambda self, mn: wrap_model(model.eval())})(  # noqa: E731
        model_name="injected", **kw)
    first = make_attack(epoch=1)(x, label)
    first_ref = A.O.run_attack("mifgsm", model, x, label, epoch=1)
    flipped = int((first != first_ref).sum())
    print("configs[1] x4, iteration 1: %d of %d pixels differ from the reference's first iterate" % (flipped, first.numel()))
    assert flipped <= 1e-5 * first.numel()
    delta = make_attack()(x, label)
    assert float(delta.abs().max()) <= A.EPS + 1e-7
    out = quantize_images(x, delta)
    print("configs[1] x4, K = 10: uint8 mismatch vs the reference's golden images %.2f%%"
          % (100 * float((out != g["adv_u8"]).mean())))


@pytest.fixture
def reference_sum_order(monkeypatch):
    """TA_ATEN_SUM_LANES=8: |g| summed in the order of the AVX2 reference that wrote the goldens"""
    host_kernels.install(monkeypatch, tag="aten8", env={"TA_ATEN_SUM_LANES": "8"})
    monkeypatch.setattr(A, "DEV", "cpu")
    monkeypatch.setattr(L, "DEV", "cpu")


@pytest.mark.parametrize("name", _subset(["mifgsm", "nifgsm", "vmifgsm", "vnifgsm", "dim", "tim", "sim", "admix", "dts", "ens"],
                                         {"mifgsm", "nifgsm", "vmifgsm", "dim", "tim", "sim", "admix", "dts", "ens"}))
def test_loops_bit_exact_in_reference_sum_order(golden, reference_sum_order, name):
    """with the one remaining difference removed -- the order in which |g| is added -- the kernels' code reproduces the
    reference's golden PERTURBATIONS (fp32, not just the uint8 images) bit for bit, momentum attacks included"""
    from transferattack_amd import backbones
    g = golden("loops_toy")
    x, label = A.t(g["x_u8"]).float() / 255, A.t(g["label"])
    models = [backbones.create("toy_cnn", seed=3, verbose=False)]
    if name == "ens":
        models.append(backbones.create("toy_cnn", seed=4, verbose=False))
    torch.manual_seed(1234)
    assert np.array_equal(A.make(name, models)(x, label).numpy(), g["delta_" + name])


def test_config2_byte_identical_in_reference_sum_order(golden, reference_sum_order):
    """BASELINE.json configs[1] in miniature (MI-FGSM, ResNet-50, K = 10): end to end through the kernel sources, the
    images are the reference's golden bytes -- the 14 % of test_config2_end_to_end_through_kernels came from the last bit
    of sum|g| alone"""
    from conftest import u8_images
    from transferattack_amd import backbones
    from transferattack_amd.utils import quantize_images, wrap_model
    import transferattack_amd as ta
    g = golden("config2_mifgsm_resnet50_n4")
    x = u8_images(4, 224, int(g["seed_images"])).float() / 255
    model = backbones.create("resnet50", seed=int(g["seed_weights"]), verbose=False)
    cls = ta.load_attack_class("mifgsm")
    atk = type("HostMIFGSM", (cls,), {"load_model": lambda self, mn: wrap_model(model.eval())})(model_name="injected")
    assert np.array_equal(quantize_images(x, atk(x, A.t(g["label"]))), g["adv_u8"])


@pytest.mark.parametrize("name,kw", [("pifgsm", {}), ("emifgsm", {}), ("iefgsm", {}), ("gnp", {}), ("gra", dict(num_neighbor=5)),
                                     ("pgn", dict(num_neighbor=4)), ("gifgsm", {}), ("dta", dict(K=3)), ("pcifgsm", {}),
                                     ("smifgrm", dict(num_neighbor=4))])
def test_widened_gradient_attacks_bit_exact_in_reference_sum_order(golden, reference_sum_order, name, kw):
    g = golden("loops_more")
    base = golden("loops_toy")
    x, label = A.t(base["x_u8"]).float() / 255, A.t(base["label"])
    atk = A.make(name, **kw)
    torch.manual_seed(1234)
    assert np.array_equal(atk(x, label).numpy(), g["delta_" + name])


@pytest.mark.parametrize("name", _subset(["svre", "cwa", "adaea", "smer"], {"svre", "cwa", "adaea"}))
def test_member_ensembles_bit_exact_in_reference_sum_order(golden, reference_sum_order, name):
    from transferattack_amd import backbones
    from transferattack_amd.utils import EnsembleModel, wrap_model
    import transferattack_amd as ta
    base = golden("loops_toy")
    x, label = A.t(base["x_u8"]).float() / 255, A.t(base["label"])
    three = name in ("adaea", "smer")
    g = golden("loops_ens" if three else "loops_more")
    models = [backbones.create("toy_cnn", seed=s, verbose=False) for s in ((3, 4, 5) if three else (3, 4))]
    cls = ta.load_attack_class(name)
    atk = type("Host" + cls.__name__, (cls,), {
        "load_model": lambda self, mn: EnsembleModel([wrap_model(m.eval()) for m in models])})(
        model_name=["a", "b", "c"][:len(models)])
    atk.noise_source = (lambda shape, lo, hi: torch.randn(shape)) if name == "adaea" else (
        lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi))
    torch.manual_seed(1234)
    np.random.seed(99)
    assert np.array_equal(atk(x, label).numpy(), g["delta_" + name])


def test_sia_bit_exact_in_reference_sum_order(golden, reference_sum_order):
    g, base = golden("sia"), golden("loops_toy")
    x, label = A.t(base["x_u8"]).float() / 255, A.t(base["label"])
    atk = A.make("sia", num_scale=4)
    np.random.seed(99)
    torch.manual_seed(1234)
    assert np.array_equal(atk(x, label).numpy(), g["delta_sia"])


@pytest.mark.parametrize("tag", list(L.CASES))
def test_recorded_hook_calls(golden, tag):
    """L2 branch, random starts, tensor and negative step: every hook call of the reference's loops (loops_hooks.npz)"""
    L.test_recorded_hook_calls(golden, tag)


@pytest.mark.parametrize("tag", list(L.CASES))
def test_loop_with_replayed_gradients(golden, tag):
    L.test_loop_with_replayed_gradients(golden, tag)


def test_folded_loop_steps_aside_for_global_module_hooks(monkeypatch):
    """a process-wide module hook (torch.nn.modules.module.register_module_forward_hook) must see the calls of the wrapper, the
    preprocessing layer and its Normalize: the folded loop, which skips those modules, is then not taken"""
    from conftest import u8_images
    x = u8_images(2, 224, 3).float() / 255
    label = torch.tensor([1, 2])
    before = _hip.stats["std_form_launches"]
    A.make("mifgsm", epoch=2)(x, label)
    assert _hip.stats["std_form_launches"] == before + 2
    seen = []
    handle = torch.nn.modules.module.register_module_forward_hook(lambda mod, inp, out: seen.append(type(mod).__name__))
    try:
        A.make("mifgsm", epoch=2)(x, label)
    finally:
        handle.remove()
    assert _hip.stats["std_form_launches"] == before + 2, "the folded loop ran although a global hook was registered"
    assert "PreprocessingModel" in seen and "_Normalize" in seen
