"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz by running the REAL reference.

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py
The reference ships no tests or golden vectors; these files are the pin: outputs
of the reference's own classes (imported through oracle/ref_shim.py) on seeded inputs.  The oracle
(oracle/fgsm_oracle.py, oracle/ta_oracle.c) is checked against them by tests/test_oracle_golden.py and
the HIP path by the -m gpu tests.  Generated with torch CPU, default thread count (8 here): for
F.interpolate the multi-threaded result is the golden one (SURVEY.md section 8c').
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
from transferattack_amd import backbones  # noqa: E402  (surrogate definitions only; no product code paths)

OUT = os.path.join(ROOT, "tests", "golden")
EPS, ALPHA = 16 / 255, 1.6 / 255


def u8_images(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, 3, size, size), generator=g, dtype=torch.uint8)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def gen_update_stack():
    """Attack.get_momentum / Attack.update_delta called on a reference MIFGSM instance."""
    atk = ref_shim.make_reference_attack("mifgsm", backbones.create("toy_cnn", verbose=False))
    g = torch.Generator().manual_seed(2)
    n, size = 3, 64
    x = u8_images(n, size, 0).float() / 255
    grad = torch.randn(n, 3, size, size, generator=g) * 1e-4
    grad[torch.rand(n, 3, size, size, generator=g) < 0.01] = 0.0         # exact zeros -> sign 0
    grad[2] = 0.0                                                        # all-zero image -> NaN momentum
    mom = torch.randn(n, 3, size, size, generator=g)
    delta = (torch.randint(-10, 11, (n, 3, size, size), generator=g).float() * ALPHA).clamp(-EPS, EPS)
    delta = torch.min(torch.max(delta, 0 - x), 1 - x)
    alpha_t = torch.rand(n, 3, size, size, generator=g) * ALPHA
    out = dict(x=x, grad=grad, momentum=mom, delta=delta, alpha_t=alpha_t, eps=EPS, alpha=ALPHA)
    for tag, decay, m_in in (("first", 1.0, 0), ("d1", 1.0, mom), ("d09", 0.9, mom), ("d0", 0.0, mom)):
        atk.decay = decay
        m_new = atk.get_momentum(grad, m_in)
        out["m_" + tag] = m_new
        out["delta_" + tag] = atk.update_delta(delta.clone().requires_grad_(True), x, m_new, ALPHA).detach()
    atk.decay = 1.0
    m_new = atk.get_momentum(grad, mom)
    out["delta_alpha_t"] = atk.update_delta(delta.clone(), x, m_new, alpha_t).detach()       # gra.py:149
    out["delta_alpha_neg"] = atk.update_delta(delta.clone(), x, m_new, -ALPHA).detach()      # cwa.py:69
    atk.norm = "l2"
    out["delta_l2"] = atk.update_delta(delta.clone(), x, grad + 1e-5, ALPHA).detach()        # attack.py:148-151
    from transferattack.utils import save_images  # noqa: F401  (quantiser arithmetic, utils.py:64)
    adv = x + out["delta_d1"]
    out["u8_d1"] = (adv.detach().permute((0, 2, 3, 1)).cpu().numpy() * 255).astype(np.uint8)
    save("update_stack", **out)


def gen_tim():
    atk = ref_shim.make_reference_attack("tim", backbones.create("toy_cnn", verbose=False))
    out = {}
    for kind in ("gaussian", "uniform", "linear"):
        out["kernel_" + kind] = atk.generate_kernel(kind, 15)
    out["kernel_gaussian_7"] = atk.generate_kernel("gaussian", 7)
    g = torch.Generator().manual_seed(5)
    c = torch.randn(2, 3, 64, 64, generator=g)
    delta = torch.zeros_like(c, requires_grad=True)
    out["grad_in"] = c
    out["grad_out"] = atk.get_grad((delta * c).sum(), delta)             # tim.py:68-74
    save("tim", **out)


def gen_dim():
    atk = ref_shim.make_reference_attack("dim", backbones.create("toy_cnn", verbose=False))
    size = 128                                                           # resize = int(128*1.1) = 140
    x = u8_images(1, size, 7).float() / 255
    gy = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(8))
    ys, gxs, seeds = [], [], []
    seed = 100
    while len(ys) < 4:                                                   # keep 3 transformed + 1 identity
        torch.manual_seed(seed)
        xin = x.clone().requires_grad_(True)
        y = atk.transform(xin)
        identity = y is xin
        if identity and any(s[1] for s in seeds):
            seed += 1
            continue
        gx = gy.clone() if identity else torch.autograd.grad(y, xin, gy)[0]
        ys.append(y.detach())
        gxs.append(gx)
        seeds.append((seed, identity))
        seed += 1
    save("dim", x=x, gy=gy, y=torch.stack(ys), gx=torch.stack(gxs), seeds=np.array([s[0] for s in seeds]),
         identity=np.array([s[1] for s in seeds]), resize_rate=1.1, diversity_prob=0.5)


def gen_copies():
    sim = ref_shim.make_reference_attack("sim", backbones.create("toy_cnn", verbose=False))
    adm = ref_shim.make_reference_attack("admix", backbones.create("toy_cnn", verbose=False))
    n, size = 4, 32
    x = u8_images(n, size, 9).float() / 255
    g = torch.Generator().manual_seed(10)
    xin = x.clone().requires_grad_(True)
    y = sim.transform(xin)
    gy = torch.randn(y.shape, generator=g)
    out = dict(x=x, sim_y=y.detach(), sim_gy=gy, sim_gx=torch.autograd.grad(y, xin, gy)[0])
    torch.manual_seed(11)
    xin = x.clone().requires_grad_(True)
    y = adm.transform(xin)
    gy = torch.randn(y.shape, generator=g)
    out.update(admix_seed=11, admix_y=y.detach(), admix_gy=gy, admix_gx=torch.autograd.grad(y, xin, gy)[0])
    save("copies", **out)


def _dts_class():
    """DTS composition exactly as SURVEY.md a17: the reference's own DIM/TIM/SIM methods."""
    ta = ref_shim.import_reference()
    TIM = ta.load_attack_class("tim")
    DIM = ta.load_attack_class("dim")
    SIM = ta.load_attack_class("sim")

    class DTS(TIM):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.resize_rate, self.diversity_prob, self.num_scale = 1.1, 0.5, 5

        def transform(self, x, **kw):
            return DIM.transform(self, SIM.transform(self, x))

        get_loss = SIM.get_loss

    return DTS


def gen_loops():
    """Whole K-iteration attacks by the reference's classes on a toy CNN (4 images 3x32x32)."""
    n, size = 4, 32
    x = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    out = dict(x_u8=u8_images(n, size, 20), label=label)
    def record_grads(atk):
        """per-iteration output of the reference's own get_grad (after TIM smoothing where it applies)"""
        grads = []
        orig = atk.get_grad

        def get_grad(loss, delta, **kw):
            grads.append(orig(loss, delta, **kw).clone())
            return grads[-1]

        atk.get_grad = get_grad
        return grads

    traced = ("mifgsm", "nifgsm", "dim", "tim", "sim", "admix")
    for name in ("fgsm", "ifgsm", "mifgsm", "nifgsm", "vmifgsm", "vnifgsm", "dim", "tim", "sim", "admix"):
        atk = ref_shim.make_reference_attack(name, backbones.create("toy_cnn", seed=3, verbose=False))
        grads = record_grads(atk) if name in traced else None
        torch.manual_seed(1234)
        out["delta_" + name] = atk(x, label)
        if grads is not None:
            out["grads_" + name] = torch.stack(grads)
    atk = ref_shim.make_reference_attack(
        "ens", [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)])
    torch.manual_seed(1234)
    out["delta_ens"] = atk(x, label)
    backbone = backbones.create("toy_cnn", seed=3, verbose=False)
    from transferattack.utils import wrap_model
    DTS = _dts_class()
    DTS.load_model = lambda self, name: wrap_model(backbone.eval())
    dts = DTS(model_name="injected")
    grads = record_grads(dts)
    torch.manual_seed(1234)
    out["delta_dts"] = dts(x, label)
    out["grads_dts"] = torch.stack(grads)
    # targeted + random_start variants of MI-FGSM
    atk = ref_shim.make_reference_attack("mifgsm", backbones.create("toy_cnn", seed=3, verbose=False), targeted=True)
    tgt = (label + 1) % 10
    out["target"] = tgt
    out["delta_mifgsm_targeted"] = atk(x, [label, tgt])
    atk = ref_shim.make_reference_attack("mifgsm", backbones.create("toy_cnn", seed=3, verbose=False), random_start=True)
    torch.manual_seed(77)
    out["delta_mifgsm_random_start"] = atk(x, label)
    save("loops_toy", **out)


def _record_hooks(atk, rec):
    """wrap the reference instance's own hooks (attack.py:118-153) so that every call of the loop leaves its inputs and
    outputs in ``rec`` in call order: (hook, {argument: tensor / float})"""
    def clone(v):
        return v.detach().clone() if torch.is_tensor(v) else v

    for hook in ("init_delta", "get_grad", "get_momentum", "update_delta"):
        orig = getattr(atk, hook)

        def wrapped(*args, _orig=orig, _hook=hook, **kw):
            out = _orig(*args, **kw)
            if _hook == "init_delta":
                rec.append((_hook, dict(out=clone(out))))
            elif _hook == "get_grad":
                rec.append((_hook, dict(out=clone(out))))
            elif _hook == "get_momentum":
                rec.append((_hook, dict(grad=clone(args[0]), momentum=clone(args[1]), out=clone(out))))
            else:
                rec.append((_hook, dict(delta=clone(args[0]), grad=clone(args[2]), alpha=clone(args[3]), out=clone(out))))
            return out
        setattr(atk, hook, wrapped)


def gen_loops_hooks():
    """Loop-level goldens for the branches of the base hooks that the plain L-inf / scalar-alpha loops never take: the REAL
    reference runs whole attacks on the toy CNN and every hook call of the loop is recorded (inputs and outputs, in order):
      * MI-FGSM with norm='l2' (update_delta's L2 branch, attack.py:148-151), zero start and random start (init_delta's
        L2 branch, attack.py:136-140);
      * MI-FGSM with random_start on the L-inf ball (attack.py:133-134);
      * GRA -- update_delta with a TENSOR step M * alpha (gra.py:149);
      * CWA on two members -- update_delta with a NEGATIVE step (cwa.py:69), random start by default.
    tests/test_hip_loops_golden.py replays them on MI355X."""
    n, size = 2, 32
    xu8 = u8_images(n, size, 20)
    x = xu8.float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    out = dict(x_u8=xu8, label=label, seed=1234)
    toy = lambda seed=3: backbones.create("toy_cnn", seed=seed, verbose=False)        # noqa: E731
    cases = (
        ("mifgsm_l2", "mifgsm", dict(norm="l2", epsilon=3.0, alpha=0.6), False),
        ("mifgsm_l2_random", "mifgsm", dict(norm="l2", epsilon=3.0, alpha=0.6, random_start=True), False),
        ("mifgsm_linf_random", "mifgsm", dict(random_start=True), False),
        ("gra", "gra", dict(num_neighbor=2, epoch=5), False),
        ("cwa", "cwa", dict(epoch=4), True),
    )
    for tag, name, kw, ens in cases:
        if ens:
            members = [toy(3), toy(4)]
            ta = ref_shim.import_reference()
            from transferattack.utils import wrap_model, EnsembleModel
            base = ta.load_attack_class(name)
            cls = type("Ref_" + name, (base,), {"load_model": lambda self, mn: EnsembleModel([wrap_model(m.eval()) for m in members])})
            atk = cls(model_name=["a", "b"], **kw)
        else:
            atk = ref_shim.make_reference_attack(name, toy(), **kw)
        rec = []
        _record_hooks(atk, rec)
        torch.manual_seed(1234)
        delta = atk(x, label)
        out[tag + ".delta"] = delta
        out[tag + ".hooks"] = np.array([h for h, _ in rec])
        seen = {}                                    # a tensor that is bit-identical to an earlier one is stored as its key
        for k, (hook, args) in enumerate(rec):
            for a, v in args.items():
                key = "%s.%d.%s" % (tag, k, a)
                if torch.is_tensor(v):
                    h = v.numpy().tobytes()
                    if h in seen:
                        out[key] = np.array(seen[h])
                    else:
                        seen[h] = key
                        out[key] = v
                elif a == "alpha":
                    out[key] = np.float64(v)
                # a Python 0 / 0. momentum is simply absent
        print(tag, "hook calls:", {h: sum(1 for hh, _ in rec if hh == h) for h in ("init_delta", "get_grad", "get_momentum", "update_delta")})
    save("loops_hooks", **out)


def gen_loops_more():
    """further gradient-family attacks riding on the same kernels -- whole loops by the
    reference's own classes on the toy CNN (same inputs / seeds as gen_loops)."""
    ref_shim.neutralise_cuda_calls()                       # pifgsm.py:52 builds its kernel with .cuda()
    n, size = 4, 32
    x = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    out = {}
    for name, kw in (("pifgsm", {}), ("emifgsm", {}), ("iefgsm", {}), ("gnp", {}), ("gra", dict(num_neighbor=5)),
                     ("pgn", dict(num_neighbor=4)), ("gifgsm", {}), ("dta", dict(K=3)), ("pcifgsm", {}), ("smifgrm", dict(num_neighbor=4))):
        atk = ref_shim.make_reference_attack(name, backbones.create("toy_cnn", seed=3, verbose=False), **kw)
        torch.manual_seed(1234)
        out["delta_" + name] = atk(x, label)
    atk = ref_shim.make_reference_attack("pifgsm", backbones.create("toy_cnn", seed=3, verbose=False), decay=1.0)
    out["delta_mpifgsm"] = atk(x, label)
    # per-member ensemble attacks on EnsembleModel.models[k] (random start: CPU generator;
    # SVRE's member choice: numpy generator)
    for name in ("svre", "cwa"):
        members = [backbones.create("toy_cnn", seed=3, verbose=False), backbones.create("toy_cnn", seed=4, verbose=False)]
        ta = ref_shim.import_reference()
        from transferattack.utils import wrap_model, EnsembleModel
        base = ta.load_attack_class(name)
        cls = type("Ref_" + name, (base,), {"load_model": lambda self, mn: EnsembleModel([wrap_model(m.eval()) for m in members])})
        atk = cls(model_name=["a", "b"])
        torch.manual_seed(1234)
        np.random.seed(99)
        out["delta_" + name] = atk(x, label)
    save("loops_more", **out)


TAIL = (("mig", dict(s_factor=5)), ("aifgtm", {}), ("mef", dict(num_neighbor=4, epoch=6)), ("gaa", dict(N=3, epoch=5)),
        ("dem", {}))


def gen_loops_tail():
    """the long tail of SURVEY.md 2.2 that rides on the same hooks: MIG (mig.py:41-84), AI-FGTM (aifgtm.py:53-95), MEF
    (mef.py:69-128), GAA (gaa.py:44-100), DEM (dem.py:52-117) -- whole loops by the reference's own classes on the toy CNN"""
    n, size = 4, 32
    x = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    out = {}
    for name, kw in TAIL:
        atk = ref_shim.make_reference_attack(name, backbones.create("toy_cnn", seed=3, verbose=False), **kw)
        torch.manual_seed(1234)
        out["delta_" + name] = atk(x, label)
    save("loops_tail", **out)


TAIL2 = (("ifgssm", {}), ("vaifgsm", dict(epoch=4)), ("adamsi_fgm", {}),
         ("rgmifgsm", dict(num_directions=2, pre_epoch=2, epoch=4)), ("dual_mifgsm", dict(epoch=5)),
         ("ens_mifgsm", dict(epoch=3, num_d=2)), ("maskblock", dict(patch_size=16)), ("usmm", dict(num_scale=3, num_mix=2)),
         ("anda", dict(n_ens=4, epoch=3)),
         ("rap", dict(epoch=6, transpoint=3, adv_steps=2)), ("decowa", dict(num_warping=3, epoch=3)),
         ("foolmix", dict(epoch=4, m=3, n=2, k=3, grad_chunk_size=5, print_timing=False)),
         ("ops", dict(num_sample_neighbor=2, num_sample_operator=3, epoch=2)))


def gen_loops_tail2():
    """more of SURVEY.md 2.2 / 2.3 on the same hooks, whole loops by the reference's own classes on the toy CNN: I-FGS2M
    (ifgssm.py:32-63), VA-I-FGSM (vaifgsm.py:31-126; its class count set to the toy's 10), AdaMSI-FGM
    (adamsi_fgm.py:31-82), the MI-FGSM tricks (mifgsm_with_tricks.py:15-266), MaskBlock (maskblock.py:34-57), US-MM
    (usmm.py:34-99), ANDA (anda.py:45-210, one image)"""
    ref_shim.neutralise_cuda_calls()
    n, size = 4, 32
    x = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    out = {}
    for name, kw in TAIL2:
        atk = ref_shim.make_reference_attack(name, backbones.create("toy_cnn", seed=3, verbose=False), **kw)
        if name == "vaifgsm":
            atk.num_classes = 10
        import random
        random.seed(11); np.random.seed(11); torch.manual_seed(1234)
        first = 1 if name == "anda" else n
        out["delta_" + name] = atk(x[:first], label[:first]).detach()
    # SSM with tricks (ssm_with_tricks.py:17-470): the Gaussian is hard-coded 3 x 224 x 224 -> 224-pixel input
    x224 = u8_images(1, 224, 23).float() / 255
    ref_shim.import_reference()
    import importlib
    from transferattack.utils import wrap_model
    tricks = importlib.import_module("transferattack.input_transformation.ssm_with_tricks")
    for name, kw in (("ssm_h", dict(num_spectrum=2, epoch=2)), ("ssm_p", dict(num_scale=4, epoch=3))):
        # (the reference's zoo entry for ssm_p names a class that does not exist, __init__.py:62 -> take the classes directly)
        base = {"ssm_h": tricks.SSM_H, "ssm_p": tricks.SSM_P}[name]
        toy = backbones.create("toy_cnn", seed=3, verbose=False)
        atk = type("Ref_" + name, (base,), {"load_model": lambda self, mn: wrap_model(toy.eval())})(model_name="injected", **kw)
        np.random.seed(7)
        torch.manual_seed(4321)
        out["delta_" + name] = atk(x224, label[:1]).detach()
    # L2T (l2t.py:415-529): its crops and spectrum views are hard-coded to 224 pixels
    import random
    atk = ref_shim.make_reference_attack("l2t", backbones.create("toy_cnn", seed=3, verbose=False), num_scale=2, epoch=3)
    random.seed(13); np.random.seed(13); torch.manual_seed(1313)
    out["delta_l2t"] = atk(x224, label[:1]).detach()
    # SU (su.py:39-182): 224-pixel inputs (its crop and DI sizes are the module constants), targeted by default; the
    # feature layer is resolved by surrogate NAME there -- for the toy surrogate it is the output of its third convolution
    su_cls = ref_shim.import_reference().load_attack_class("su")
    toy = backbones.create("toy_cnn", seed=3, verbose=False)
    from transferattack.utils import wrap_model as ref_wrap
    atk = type("Ref_SU", (su_cls,), {"load_model": lambda self, mn: ref_wrap(toy.eval()),
                                     "_target_layer": lambda self, mn, depth: self.model[1].body[4]})(model_name="injected", epoch=3)
    x2 = u8_images(2, 224, 29).float() / 255
    tgt = (label[:2] + 3) % 10
    random.seed(17); np.random.seed(17); torch.manual_seed(1717)
    out["delta_su"] = atk(x2, [label[:2], tgt]).detach()
    out["su_target"] = tgt
    # Everywhere Attack (everywhere.py:14-412): always targeted, labels = [truth, target]
    ev_cls = ref_shim.import_reference().load_attack_class("everywhere")
    toy = backbones.create("toy_cnn", seed=3, verbose=False)
    atk = type("Ref_Everywhere", (ev_cls,), {"load_model": lambda self, mn: ref_wrap(toy.eval())})(model_name="injected", targeted=True, epoch=8)
    random.seed(19); np.random.seed(19); torch.manual_seed(1919)
    out["delta_everywhere"] = atk(x2, [label[:2], tgt]).detach()
    save("loops_tail2", **out)


def gen_loops_ens():
    """SURVEY.md 8(f) rank 4, continued: AdaEA and SMER by the reference's own classes on three toy members
    (Gaussian / uniform start from the CPU generator; SMER's member order from the numpy generator).  SMER's member
    weights persist on the attack object, so a second batch is recorded too."""
    ref_shim.neutralise_cuda_calls()                       # smer.py:45,58-59,69 use .cuda()
    n, size = 4, 32
    x = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    x2 = u8_images(n, size, 22).float() / 255
    out = dict(x2_u8=u8_images(n, size, 22))
    ta = ref_shim.import_reference()
    from transferattack.utils import wrap_model, EnsembleModel
    for name in ("adaea", "smer"):
        members = [backbones.create("toy_cnn", seed=s, verbose=False) for s in (3, 4, 5)]
        base = ta.load_attack_class(name)
        cls = type("Ref_" + name, (base,), {"load_model": lambda self, mn: EnsembleModel([wrap_model(m.eval()) for m in members])})
        atk = cls(model_name=["a", "b", "c"])
        torch.manual_seed(1234)
        np.random.seed(99)
        out["delta_" + name] = atk(x, label)
        out["delta2_" + name] = atk(x2, label)
        if name == "smer":
            out["smer_weight"] = atk.weight_selection.weight.detach().clone()
    # rank 3, last member: FGSRA (DCT-domain neighbours; tensor step).  rand_like draws come from the CPU generator.
    atk = ref_shim.make_reference_attack("fgsra", backbones.create("toy_cnn", seed=3, verbose=False), max_iter=4)
    torch.manual_seed(1234)
    out["delta_fgsra"] = atk(x, label)
    probe = torch.rand(2, 3, 7, 10, generator=torch.Generator().manual_seed(5))
    out["dct_probe"], out["dct_2d"], out["idct_2d"] = probe, atk.dct_2d(probe), atk.idct_2d(probe)
    save("loops_ens", **out)


def gen_sia():
    """SIA's block transform (input_transformation/sia.py:86-100) by the reference's own class --
    the 20-copy stack of one seeded batch, the gradient autograd sends back through it, and a whole K=10 loop."""
    ref_shim.neutralise_cuda_calls()
    gen = torch.Generator().manual_seed(31)
    x = torch.rand(2, 3, 32, 40, generator=gen)
    atk = ref_shim.make_reference_attack("sia", backbones.create("toy_cnn", seed=3, verbose=False))
    np.random.seed(7)
    torch.manual_seed(8)
    xin = x.clone().requires_grad_(True)
    y = atk.transform(xin)
    gy = torch.randn(y.shape, generator=gen)
    gx = torch.autograd.grad(y, xin, gy)[0]
    out = dict(x=x, y=y.detach(), gy=gy, gx=gx, np_seed=7, torch_seed=8)
    n, size = 4, 32
    xl = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    atk = ref_shim.make_reference_attack("sia", backbones.create("toy_cnn", seed=3, verbose=False), num_scale=4)
    np.random.seed(99)
    torch.manual_seed(1234)
    out["delta_sia"] = atk(xl, label)
    # SSM (ssm.py:40-99): the Gaussian is hard-coded 3 x 224 x 224, so the loop runs on 224-pixel inputs
    x224 = u8_images(1, 224, 23).float() / 255
    atk = ref_shim.make_reference_attack("ssm", backbones.create("toy_cnn", seed=3, verbose=False), num_spectrum=3, epoch=3)
    torch.manual_seed(4321)
    out["delta_ssm"] = atk(x224, label[:1])
    save("sia", **out)


def gen_bsr():
    """BSR.transform (bsr.py:41-67) by the reference's own class: the stack of shuffled-and-rotated copies, its backward
    (autograd), and a whole loop on the toy CNN.  Three host generators feed it (python ``random``, numpy, torch): all
    seeded.  The rotation inside is torchvision's RandomRotation, restated in oracle/fgsm_oracle.py::rotate_tensor
    (torchvision is not installed); everything else is the reference's code."""
    import random
    atk = ref_shim.make_reference_attack("bsr", backbones.create("toy_cnn", seed=3, verbose=False), num_scale=4)
    g = torch.Generator().manual_seed(31)
    x = torch.rand(2, 3, 48, 64, generator=g)
    out = dict(x=x, num_scale=4, num_block=3, seed=77)
    random.seed(77); np.random.seed(77); torch.manual_seed(77)
    xin = x.clone().requires_grad_(True)
    y = atk.transform(xin)
    gy = torch.randn(y.shape, generator=g)
    out.update(y=y.detach(), gy=gy, gx=torch.autograd.grad(y, xin, gy)[0])
    n, size = 4, 32
    xl = u8_images(n, size, 20).float() / 255
    label = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(21))
    atk = ref_shim.make_reference_attack("bsr", backbones.create("toy_cnn", seed=3, verbose=False), num_scale=5)
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    out["delta_bsr"] = atk(xl, label)
    out["loop_scale"] = 5
    save("bsr", **out)


def gen_config1():
    """BASELINE.json configs[0]: I-FGSM on ResNet-18, 16 images, eps=16/255, K=10, CPU reference path."""
    n = 16
    xu8 = u8_images(n, 224, 0)
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1))
    atk = ref_shim.make_reference_attack("ifgsm", backbones.create("resnet18", seed=0, verbose=False))
    x = xu8.float() / 255
    delta = atk(x, label)
    adv_u8 = ((x + delta).permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)          # utils.py:64
    import zlib
    save("config1_ifgsm_resnet18", x_crc32=zlib.crc32(xu8.numpy().tobytes()), label=label, adv_u8=adv_u8,
         seed_images=0, seed_labels=1, seed_weights=0)          # x_u8 = u8_images(16, 224, seed_images)


def gen_config2():
    """BASELINE.json configs[1] in miniature: MI-FGSM on ResNet-50 (seeded init, calibrated BatchNorm), 4 of the
    synthetic images, eps=16/255, alpha=1.6/255, K=10, by the reference's own class on the CPU."""
    n = 4
    xu8 = u8_images(n, 224, 0)
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1))
    atk = ref_shim.make_reference_attack("mifgsm", backbones.create("resnet50", seed=0, verbose=False))
    x = xu8.float() / 255
    delta = atk(x, label)
    adv_u8 = ((x + delta).permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)          # utils.py:64
    save("config2_mifgsm_resnet50_n4", label=label, adv_u8=adv_u8, seed_images=0, seed_labels=1, seed_weights=0)


def _adv_u8(x, delta):
    return ((x + delta).permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)            # utils.py:64


def gen_config3():
    """BASELINE.json configs[2] in miniature: DTS (DIM o SIM on the input, TIM on the gradient) on
    ResNet-50, 2 images x 5 scale copies, K=10, by the reference's own DIM / TIM / SIM methods on the CPU."""
    n = 2
    xu8 = u8_images(n, 224, 0)
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1))
    backbone = backbones.create("resnet50", seed=0, verbose=False)
    DTS = _dts_class()                                     # imports the reference package first
    from transferattack.utils import wrap_model
    DTS.load_model = lambda self, name: wrap_model(backbone.eval())
    atk = DTS(model_name="injected")
    x = xu8.float() / 255
    torch.manual_seed(1234)
    delta = atk(x, label)
    save("config3_dts_resnet50_n2", label=label, adv_u8=_adv_u8(x, delta), seed_images=0, seed_labels=1, seed_weights=0,
         seed_draws=1234)


def gen_config4():
    """BASELINE.json configs[3] in miniature: VMI-FGSM on ViT-B/16 (timm layout, seeded init), 2 images, 4 variance
    samples (20 in the config), K=3 (10 in the config), by the reference's own class on the CPU
    (gradient/vmifgsm.py:42-97); and the same on ResNet-18 with the full 20 samples, K=3."""
    n = 2
    xu8 = u8_images(n, 224, 0)
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1))
    x = xu8.float() / 255
    out = dict(label=label, seed_images=0, seed_labels=1, seed_weights=0, seed_draws=1234)
    for tag, model, kw in (("vit", "vit_base_patch16_224", dict(num_neighbor=4, epoch=3)),
                           ("resnet18", "resnet18", dict(num_neighbor=20, epoch=3))):
        atk = ref_shim.make_reference_attack("vmifgsm", backbones.create(model, seed=0, verbose=False), **kw)
        torch.manual_seed(1234)
        out["adv_u8_" + tag] = _adv_u8(x, atk(x, label))
    save("config4_vmifgsm_n2", **out)


def gen_config5():
    """BASELINE.json configs[4] in miniature: ensemble MI-FGSM over ResNet-50 + VGG-16 + Inception-v3 (299-pixel branch of
    wrap_model) + ViT-B/16 through the reference's EnsembleModel (utils.py:82-105), 2 images, K=3."""
    n = 2
    xu8 = u8_images(n, 224, 0)
    label = torch.randint(0, 1000, (n,), generator=torch.Generator().manual_seed(1))
    members = [backbones.create(m, seed=0, verbose=False) for m in ENS_MEMBERS]
    atk = ref_shim.make_reference_attack("ens", members, epoch=3)
    x = xu8.float() / 255
    delta = atk(x, label)
    save("config5_ens4_n2", label=label, adv_u8=_adv_u8(x, delta), seed_images=0, seed_labels=1, seed_weights=0)


ENS_MEMBERS = ("resnet50", "vgg16", "inception_v3", "vit_base_patch16_224")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["update_stack", "tim", "dim", "copies", "loops", "loops_hooks", "loops_more", "loops_tail", "loops_tail2", "loops_ens", "sia", "bsr", "config1", "config2", "config3", "config4", "config5"]
    for w in which:
        globals()["gen_" + w]()
