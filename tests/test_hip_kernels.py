"""GPU (-m gpu): the HIP kernels, called through the C-ABI (ctypes binding), against
  (a) the golden vectors produced by the REAL reference (tests/golden, oracle/gen_golden.py),
  (b) the plain-C oracle (oracle/ta_oracle.c) and the torch-CPU oracle (oracle/fgsm_oracle.py) on seeded inputs.

Tolerances (written out, per BASELINE.json north_star: fp32 within 1e-5, final uint8 bit-exact):
  * delta / uint8 / TIM / DIM fwd+bwd / SIM / Admix / quantiser / Philox stream: BIT-EXACT.
  * momentum: the per-image sum|g| is added in a different (fixed) order than ATen's AVX2 cascade, so the
    mean can differ in its last bits -> m' = m*decay + g/mean within 8 * 2^-24 * (|m*decay| + |g/mean|)
    (conftest.assert_momentum_close); the *sign* (all that reaches delta) can then differ only where |m'| is
    itself within rounding of zero: such elements must have |m'| < 1e-5 and be < 1e-6 of all elements.
  * DIM forward vs the torch op: <= 2 ulp (ATen's own result depends on its work partitioning); vs the C
    oracle (same recipe): bit-exact.
"""
import numpy as np
import pytest
import torch

import c_oracle as C
import fgsm_oracle as O
from conftest import assert_momentum_close, ulp_diff
from transferattack_amd import _hip

pytestmark = pytest.mark.gpu
EPS, ALPHA = 16 / 255, 1.6 / 255
DEV = "cuda"


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV).contiguous()


def host(t):
    return t.detach().cpu().numpy()


def assert_delta_equal(d_hip, d_ref, m_ref):
    """bit-exact, except where the reference momentum is within rounding of zero (see module docstring)."""
    d_hip, d_ref, m_ref = np.asarray(d_hip), np.asarray(d_ref), np.asarray(m_ref)
    bad = d_hip != d_ref
    if bad.any():
        assert (np.abs(m_ref[bad]) < 1e-5).all(), "delta differs where the momentum sign is unambiguous"
        assert bad.sum() <= max(1, int(1e-6 * bad.size)), "too many sign flips: %d" % bad.sum()


# ------------------------------------------------------------------------------------------ update stack
@pytest.mark.parametrize("tag,decay,first", [("first", 1.0, True), ("d1", 1.0, False), ("d09", 0.9, False),
                                             ("d0", 0.0, False)])
def test_update_stack_golden(golden, tag, decay, first):
    g = golden("update_stack")
    grad, mom, delta, x = dev(g["grad"]), dev(g["momentum"]), dev(g["delta"]), dev(g["x"])
    # hook-level kernels
    m_out = torch.empty_like(grad)
    _hip.momentum(grad, None if first else mom, m_out, decay)
    assert_momentum_close(host(m_out), g["m_" + tag], g["grad"], None if first else g["momentum"], decay)
    assert np.isnan(host(m_out)[2]).all()                               # zero-gradient image: NaN momentum
    d_out = torch.empty_like(delta)
    _hip.update_delta_linf(delta, x, dev(g["m_" + tag]), ALPHA, EPS, d_out)
    assert np.array_equal(host(d_out), g["delta_" + tag])               # same momentum in -> same bytes out
    # fused path: the same as the two hooks, and as the reference
    d = delta.clone()
    m = torch.empty_like(grad)
    _hip.mi_update(grad, None if first else mom.clone(), m, d, x, decay, ALPHA, EPS)
    assert_momentum_close(host(m), g["m_" + tag], g["grad"], None if first else g["momentum"], decay)
    assert_delta_equal(host(d), g["delta_" + tag], g["m_" + tag])
    assert np.array_equal(host(d)[2], g["delta"][2])                    # NaN momentum -> frozen delta
    assert np.array_equal(host(m), host(m_out), equal_nan=True)


def test_update_delta_variants_golden(golden):
    g = golden("update_stack")
    delta, x, m = dev(g["delta"]), dev(g["x"]), dev(g["m_d1"])
    out = torch.empty_like(delta)
    _hip.update_delta_linf(delta, x, m, dev(g["alpha_t"]), EPS, out)            # tensor step (gra.py:149)
    assert np.array_equal(host(out), g["delta_alpha_t"])
    _hip.update_delta_linf(delta, x, m, -ALPHA, EPS, out)                       # negative step (cwa.py:69)
    assert np.array_equal(host(out), g["delta_alpha_neg"])
    xadv = torch.empty_like(delta)
    _hip.update_delta_linf(delta, x, m, ALPHA, EPS, out, x_adv=xadv)
    assert np.array_equal(host(xadv), g["x"] + g["delta_d1"])
    _hip.update_delta_l2(delta, x, dev(g["grad"] + np.float32(1e-5)), ALPHA, EPS, out)     # attack.py:148-151
    np.testing.assert_allclose(host(out), g["delta_l2"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("shape", [(32, 3, 224, 224), (5, 3, 37, 41), (1, 3, 8, 8), (3, 3, 299, 299), (2, 1, 1, 7)])
def test_fused_update_random(shape):
    """BASELINE sizes and ragged ones (E not a multiple of 4, E < one tile, N = 1) against the torch oracle."""
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randint(0, 256, shape, generator=gen).float() / 255
    grad = torch.randn(shape, generator=gen) * 1e-4
    grad[torch.rand(shape, generator=gen) < 0.01] = 0
    mom = torch.randn(shape, generator=gen)
    var = torch.randn(shape, generator=gen) * 1e-5
    delta = O.box_clamp((torch.randint(-10, 11, shape, generator=gen).float() * ALPHA).clamp(-EPS, EPS), 0 - x, 1 - x)
    for decay, use_var in ((1.0, False), (0.9, True)):
        gsum = grad + var if use_var else grad
        m_ref = O.momentum_step(gsum, mom, decay)
        d_ref = O.delta_step(delta, x, m_ref, ALPHA, EPS)
        d, m, xa = delta.clone().to(DEV), mom.clone().to(DEV), torch.empty(shape, device=DEV)   # updated in place
        _hip.mi_update(grad.to(DEV), m, m, d, x.to(DEV), decay, ALPHA, EPS,
                       variance=var.to(DEV) if use_var else None, x_adv=xa)
        assert_momentum_close(host(m), m_ref.numpy(), gsum.numpy(), mom.numpy(), decay)
        assert_delta_equal(host(d), d_ref.numpy(), m_ref.numpy())
        assert np.array_equal(host(xa), host(x.to(DEV) + d))
        # plain-C oracle says the same
        assert_delta_equal(host(d), C.update_delta_linf(delta.numpy(), x.numpy(), m_ref.numpy(), ALPHA, EPS),
                           m_ref.numpy())


def test_update_without_momentum():
    """decay == 0 (FGSM / I-FGSM): m' = m*0 + g/mean|g| never depends on a finite old momentum, so the fused update
    neither reads nor stores it (16 B/element) -- same delta as the full form, for every iteration of a replayed loop"""
    gen = torch.Generator().manual_seed(11)
    for shape in ((4, 3, 224, 224), (3, 3, 37, 41)):
        x = (torch.randint(0, 256, shape, generator=gen).float() / 255).to(DEV)
        d_full, d_lean = torch.zeros(shape, device=DEV), torch.zeros(shape, device=DEV)
        m_full = None
        for it in range(4):
            grad = (torch.randn(shape, generator=gen) * 1e-4).to(DEV)
            m_new = torch.empty(shape, device=DEV)
            _hip.mi_update(grad, m_full, m_new, d_full, x, 0.0, ALPHA, EPS)
            m_full = m_new
            xa = torch.empty(shape, device=DEV)
            _hip.mi_update(grad, None, None, d_lean, x, 0.0, ALPHA, EPS, x_adv=xa)
            assert torch.equal(d_full, d_lean) and torch.equal(xa, x + d_lean)


def test_byte_source_of_the_fused_update():
    """ta_u8_source_probe + ta_mi_update_u8: PNG-decoded images are float(byte) / 255 (utils.py:136); the fused update may then
    read the byte (1 B instead of 4 B per element) and must produce the SAME momentum, delta and x + delta, bit for bit, in
    every instantiation (first iteration, no momentum kept, variance term, with / without x_adv, cached and streaming loads).
    A batch that is not byte-valued raises the device-side flag and the same launch reads the fp32 operand."""
    every = (torch.arange(256, dtype=torch.float32) / 255).to(DEV).contiguous()          # the IEEE quotients
    u8, flag = _hip.u8_source_probe(every)
    assert int(flag.item()) == 0 and torch.equal(u8.cpu(), torch.arange(256, dtype=torch.uint8))
    gen = torch.Generator().manual_seed(23)
    for shape in ((2, 3, 224, 224), (125, 3, 224, 224), (3, 1, 6, 6)):                     # the middle one streams (NT)
        if DEV == "cpu" and shape[0] > 8:
            continue
        xb = torch.randint(0, 256, shape, generator=gen, dtype=torch.uint8)
        x = (xb.float() / 255).to(DEV)
        src = _hip.u8_source_probe(x)
        assert int(src[1].item()) == 0 and torch.equal(src[0].cpu(), xb)
        grad = (torch.randn(shape, generator=gen) * 1e-4).to(DEV)
        var = (torch.randn(shape, generator=gen) * 1e-5).to(DEV)
        mom = torch.randn(shape, generator=gen).to(DEV)
        delta = O.box_clamp((torch.randint(-10, 11, shape, generator=gen).float() * ALPHA).clamp(-EPS, EPS), 0 - x.cpu(), 1 - x.cpu()).to(DEV)
        before = _hip.stats["u8_source_launches"]
        for m_in, keep, use_var, want_xadv in ((mom, True, False, True), (mom, True, False, False), (None, True, False, True),
                                               (None, False, False, True), (mom, True, True, True), (None, False, True, False)):
            out = {}
            for tag, source in (("fp32", None), ("bytes", src)):
                d = delta.clone()
                m_out = torch.empty_like(x) if keep else None
                xa = torch.full_like(x, float("nan")) if want_xadv else None
                _hip.abs_sum_partials(grad, var if use_var else None)
                _hip.mi_update(grad, None if m_in is None else m_in.clone(), m_out, d, x, 1.0 if keep else 0.0, ALPHA, EPS,
                               variance=var if use_var else None, x_adv=xa, data_u8=source)
                out[tag] = (d, m_out, xa)
            for a, b in zip(out["fp32"], out["bytes"]):
                assert (a is None and b is None) or torch.equal(a, b)
        assert _hip.stats["u8_source_launches"] == before + 6
        # not byte-valued: the flag goes up, the launch reads the floats
        y = x.clone()
        y.view(-1)[y.numel() // 2] = 0.123456
        src_y = _hip.u8_source_probe(y)
        assert int(src_y[1].item()) == 1
        d1, d2 = delta.clone(), delta.clone()
        m1, m2 = torch.empty_like(x), torch.empty_like(x)
        _hip.mi_update(grad, mom, m1, d1, y, 1.0, ALPHA, EPS)
        _hip.mi_update(grad, mom, m2, d2, y, 1.0, ALPHA, EPS, data_u8=src_y)
        assert torch.equal(d1, d2) and torch.equal(m1, m2)
    for bad in (float("nan"), 1.5, -0.25, 1e-9):                                           # out of range / NaN: mismatch
        z = torch.full((1, 3, 4, 4), 0.5 if bad != bad else bad, device=DEV)
        z.view(-1)[5] = bad
        assert int(_hip.u8_source_probe(z)[1].item()) == 1


@pytest.mark.parametrize("shape", [(4, 3, 224, 224), (125, 3, 224, 224), (3, 3, 37, 41), (2, 1, 5, 7), (2, 4, 6, 6)])
def test_normalize_folded_update(shape):
    """The surrogate's Normalize (utils.py:72-79) folded into both ends of an iteration (round 5):
      ta_normalize_adv_fwd == the add of attack.py:88 (as the fused update's x_adv) followed by ta_normalize_fwd,
      ta_abs_sum_partials_std + ta_mi_update_std == ta_normalize_bwd + ta_mi_update[_u8],
    BIT FOR BIT -- y, the |g| sums, momentum and delta -- for the first iteration, the steady state and decay 0, with the fp32
    and the byte source, at sizes that take the vector / streaming / scalar forms (hw % 4 != 0: scalar)."""
    if DEV == "cpu" and shape[0] > 8:
        pytest.skip("batch-size case: device only")
    gen = torch.Generator().manual_seed(sum(shape))
    n, c = shape[0], shape[1]
    xb = torch.randint(0, 256, shape, generator=gen, dtype=torch.uint8)
    x = (xb.float() / 255).to(DEV)
    mean = torch.tensor([0.485, 0.456, 0.406, 0.5][:c]).to(DEV)
    std = torch.tensor([0.229, 0.224, 0.225, 0.25][:c]).to(DEV)
    gy = (torch.randn(shape, generator=gen) * 1e-4).to(DEV)
    gy.view(-1)[::97] = 0.0                                                 # exact zeros: sign(0) = 0
    mom = torch.randn(shape, generator=gen).to(DEV)
    delta = O.box_clamp((torch.randint(-10, 11, shape, generator=gen).float() * ALPHA).clamp(-EPS, EPS), 0 - x.cpu(), 1 - x.cpu()).to(DEV)
    src = _hip.u8_source_probe(x) if x[0].numel() % 4 == 0 else None
    # forward end
    for source in (None, src):
        xa = torch.empty_like(x)
        _hip.update_delta_linf(delta, x, torch.zeros_like(x), 0.0, EPS, torch.empty_like(x), x_adv=xa)   # xa = x + delta
        assert torch.equal(xa, x + delta)
        y_ref, y = torch.empty_like(x), torch.full_like(x, float("nan"))
        _hip.normalize_fwd(xa, y_ref, mean, std)
        _hip.normalize_adv_fwd(x, delta, y, mean, std, data_u8=source)
        assert torch.equal(y, y_ref)
        if c == 3 and x[0, 0].numel() % 4 == 0:                             # the NHWC form for a channels_last surrogate: same values
            y_cl = torch.full_like(x, float("nan"), memory_format=torch.channels_last)
            _hip.normalize_adv_fwd(x, delta, y_cl, mean, std, data_u8=source)
            assert not y_cl.is_contiguous() and torch.equal(y_cl, y_ref)
            assert torch.equal(y_cl.permute(0, 2, 3, 1).contiguous(), y_ref.permute(0, 2, 3, 1).contiguous())
    # backward end
    launches = _hip.stats["std_form_launches"]
    for m_in, keep in ((mom, True), (None, True), (None, False)):
        decay = 1.0 if keep else 0.0
        gx = torch.empty_like(gy)
        _hip.normalize_bwd(gy, gx, std)
        ws_ref, slots = _hip.partials_of(gx)
        d_ref, m_ref = delta.clone(), (torch.empty_like(x) if keep else None)
        _hip.mi_update(gx, None if m_in is None else m_in.clone(), m_ref, d_ref, x, decay, ALPHA, EPS)
        for source in (None, src):
            for handed_over in (True, False):                               # sums left by a producer / K1 inside the call
                g_in = gy.clone()
                if handed_over:
                    ws, s2 = _hip.abs_sum_partials_std(g_in, std)
                    assert s2 == slots and torch.equal(ws[:n * slots], ws_ref[:n * slots]), "sums of |gy / std| differ from ta_normalize_bwd's"
                d, m_out = delta.clone(), (torch.empty_like(x) if keep else None)
                before = _hip.stats["partials_reused"]
                _hip.mi_update(g_in, None if m_in is None else m_in.clone(), m_out, d, x, decay, ALPHA, EPS, data_u8=source, std=std)
                assert _hip.stats["partials_reused"] == before + (1 if handed_over else 0)
                assert torch.equal(d, d_ref) and (m_out is None or torch.equal(m_out, m_ref))
    assert _hip.stats["std_form_launches"] == launches + 3 * 2 * 2
    # sums attached for ONE std vector are not taken for another (or for the plain form)
    g_in = gy.clone()
    _hip.abs_sum_partials_std(g_in, std)
    before = _hip.stats["k1_passes"]
    _hip.mi_update(g_in, mom.clone(), torch.empty_like(x), delta.clone(), x, 1.0, ALPHA, EPS)
    assert _hip.stats["k1_passes"] == before + 1
    with pytest.raises(ValueError):
        _hip.mi_update(gy, None, None, delta.clone(), x, 0.0, ALPHA, EPS, std=std, x_adv=torch.empty_like(x))


@pytest.mark.parametrize("n,oh,ow", [(2, 112, 112), (1, 9, 37), (3, 16, 16)])
def test_stem_kernel_leaves_the_sums(n, oh, ow):
    """ta_stem7s2_input_grad with (std, ws): the same dx, plus per-workgroup sums of |dx / std[c]| that add up to the image's
    (fp64 evaluation, summation-order tolerance) -- and ta_mi_update_std with them moves delta as it does with K1's sums"""
    gen = torch.Generator().manual_seed(n * 100 + ow)
    w = (torch.randn(64, 3, 7, 7, generator=gen) * 0.05).to(DEV)
    dy = torch.randn(n, 64, oh, ow, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    std = torch.tensor([0.229, 0.224, 0.225]).to(DEV)
    w2 = _hip.stem7s2_prepare(w)
    plain = _hip.stem7s2_input_grad(dy, w2, torch.full((n, 3, 2 * oh, 2 * ow), float("nan"), device=DEV))
    assert _hip.partials_of(plain) is None
    dx = _hip.stem7s2_input_grad(dy, w2, torch.full((n, 3, 2 * oh, 2 * ow), float("nan"), device=DEV), std=std)
    assert torch.equal(dx, plain)
    ws, slots = _hip.partials_of(dx)
    assert slots == -(-ow // 32) * -(-oh // 4)
    got = ws[:n * slots].double().reshape(n, slots).sum(1).cpu()
    want = (dx.double() / std.double().view(1, 3, 1, 1)).abs().reshape(n, -1).sum(1).cpu()
    assert float(((got - want).abs() / want).max()) <= 1e-6
    x = (torch.randint(0, 256, dx.shape, generator=gen, dtype=torch.uint8).float() / 255).to(DEV)
    d1, d2 = torch.zeros_like(x), torch.zeros_like(x)
    m1, m2 = torch.empty_like(x), torch.empty_like(x)
    before = _hip.stats["partials_reused"]
    _hip.mi_update(dx, None, m1, d1, x, 1.0, ALPHA, EPS, std=std)            # the stem kernel's sums
    assert _hip.stats["partials_reused"] == before + 1
    _hip.mi_update(plain, None, m2, d2, x, 1.0, ALPHA, EPS, std=std)         # K1 inside the call
    q = (plain / std.view(1, 3, 1, 1)).abs()
    assert_momentum_close(host(m1), host(m2), host(plain / std.view(1, 3, 1, 1)), None, 1.0)
    assert_delta_equal(host(d1), host(d2), host(m2))


@pytest.mark.parametrize("n,in_size,out_size", [(2, 224, 299), (3, 37, 50), (1, 8, 11), (2, 64, 96), (1, 100, 101)])
def test_resize_normalize_kernels(n, in_size, out_size):
    """PreprocessingModel with a Resize (utils.py:50-53, 72-79: Inception-v3, 224 -> 299, mean = std = 0.5) as one kernel each
    way, against the ops the reference runs: F.interpolate(bilinear, align_corners=False) + Normalize on torch's CPU path and
    their autograd backward.  Forward: ATen's own result depends on its thread partitioning in the last bit (SURVEY 8c'),
    so within 2.4e-7 / std (+ 1 ulp of the result) after the Normalize; backward: bit-exact (ATen's accumulation
    order is thread-count invariant), and the |gx| tile sums equal K1's to summation order."""
    gen = torch.Generator().manual_seed(n * 1000 + in_size)
    x = torch.randint(0, 256, (n, 3, in_size, in_size), generator=gen).float() / 255
    mean, std = torch.tensor([0.5, 0.45, 0.4]), torch.tensor([0.5, 0.25, 0.2])
    xr = x.clone().requires_grad_(True)
    v = torch.nn.functional.interpolate(xr, size=(out_size, out_size), mode="bilinear", align_corners=False)
    y_ref = (v - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    gy = torch.randn(y_ref.shape, generator=gen)
    gx_ref = torch.autograd.grad(y_ref, xr, gy)[0]
    y = torch.full(y_ref.shape, float("nan"), device=DEV)
    _hip.resize_normalize_fwd(x.to(DEV), y, mean.to(DEV), std.to(DEV))
    err = (host(y).astype(np.float64) - y_ref.detach().numpy()).__abs__()
    # ATen's bilinear on [0, 1] data moves by <= 1.8e-7 with its thread partitioning (SURVEY 8c'); then one division by std
    bound = 2.4e-7 / std.view(1, 3, 1, 1).numpy() + 2.0 ** -23 * np.abs(y_ref.detach().numpy())
    assert (err <= bound).all(), "forward off by %.3e" % float(err.max())
    gx = torch.full(x.shape, float("nan"), device=DEV)
    _hip.resize_normalize_bwd(gy.to(DEV), gx, std.to(DEV))
    assert np.array_equal(host(gx), gx_ref.numpy()), "backward differs: max %.3e" % float(np.abs(host(gx) - gx_ref.numpy()).max())
    ws, slots = _hip.partials_of(gx)
    sums = host(ws)[:n * slots].reshape(n, slots).astype(np.float64).sum(1)
    want = np.abs(gx_ref.numpy().astype(np.float64)).reshape(n, -1).sum(1)
    assert np.allclose(sums, want, rtol=1e-5)


def test_launch_timing():
    """ta_timing_begin / ta_timing_end: the fused update carries HIP events on its own dispatch packets -- same results
    as an untimed call, one positive duration per call (two-launch and handed-over forms), never longer than the
    hipEventRecord markers around the same call, and the session ends cleanly when fewer calls than capacity were made"""
    gen = torch.Generator().manual_seed(5)
    shape = (32, 3, 224, 224)
    x = (torch.randint(0, 256, shape, generator=gen).float() / 255).to(DEV)
    grad = (torch.randn(shape, generator=gen) * 1e-4).to(DEV)
    m0 = torch.randn(shape, generator=gen).to(DEV)
    want_m, want_d = torch.empty_like(m0), torch.zeros(shape, device=DEV)
    _hip.mi_update(grad, m0, want_m, want_d, x, 1.0, ALPHA, EPS)
    torch.cuda.synchronize()
    _hip.timing_begin(8)
    markers, outs = [], []
    for handed_over in (False, True, False):
        got_m, got_d = torch.empty_like(m0), torch.zeros(shape, device=DEV)
        if handed_over:
            _hip.abs_sum_partials(grad)                     # registers the |g| sums: the timed call is the update kernel alone
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        _hip.mi_update(grad, m0, got_m, got_d, x, 1.0, ALPHA, EPS)
        end.record()
        markers.append((start, end))
        outs.append((got_m, got_d))
    torch.cuda.synchronize()
    ms = _hip.timing_end()
    assert len(ms) == 3
    for (got_m, got_d), (start, end), t in zip(outs, markers, ms):
        assert torch.equal(got_m, want_m) and torch.equal(got_d, want_d)
        # a duration the device can have: longer than 135 MB at twice the HBM peak, shorter than a second.  (Round 3 also
        # asserted "<= the hipEventRecord markers around the call" and "the handed-over form is not slower than the two-launch
        # form": relations between SINGLE timing samples, which tripped on a loaded box in r4b -- 39.4 vs 28.6 us -- and,
        # collected first in a -x tier, would have hidden everything behind them.  They are printed, not asserted.)
        assert 0.005 < t < 1000.0, t
    print("dispatch-clock durations of three update calls [two-launch, handed over, two-launch]: %s us; hipEventRecord markers "
          "around them: %s us" % (["%.1f" % (1e3 * t) for t in ms], ["%.1f" % (1e3 * s.elapsed_time(e)) for s, e in markers]))
    with pytest.raises(_hip.HipExtensionError):
        _hip.timing_end()                                   # not armed any more
    _hip.mi_update(grad, m0, got_m, got_d, x, 1.0, ALPHA, EPS)          # untimed launches keep working
    torch.cuda.synchronize()


def _k1_sums(t):
    """per-image sum|t| the way K1 adds it: [n] float32 (the fixed-order total the fused update divides by E)"""
    n, e = t.shape[0], t[0].numel()
    keep = getattr(t, _hip._ATTR, None)
    ws, slots = _hip.abs_sum_partials(t)
    _hip.invalidate_partials(t)
    if keep is not None:
        setattr(t, _hip._ATTR, keep)             # the K1 run above replaced the producer's sums: put them back
    return ws[:n * slots].view(n, slots)


def _check_registered(out, exact):
    """the registry holds sums for ``out``; per image they add up to sum|out| (bit-identical per-tile sums to K1's for
    the kernels that use K1's tiling, else equal to an fp64 sum within fp32 summation error)"""
    assert _hip.partials_of(out) is not None, "the producer attached no sums to its output"
    ws, slots = _hip.partials_of(out)
    n = out.shape[0]
    sums = ws[:n * slots].view(n, slots).clone()
    k1 = _k1_sums(out)
    if exact:
        assert slots == k1.shape[1] and torch.equal(sums, k1)
    truth = out.double().abs().flatten(1).sum(1)
    np.testing.assert_allclose(host(sums.double().sum(1)), host(truth), rtol=2e-6)
    return sums


def test_producer_side_partials_all_kernels():
    """every kernel that can be the LAST writer of the input gradient leaves the per-tile sums of |g|; with them the
    fused update gives what it gives after its own K1 pass (bit for bit where the tiling is K1's; else the momentum
    within the summation-order bound and delta equal wherever the momentum sign is unambiguous)"""
    gen = torch.Generator().manual_seed(5)
    for shape in ((4, 3, 224, 224), (2, 3, 37, 41)):
        n = shape[0]
        data = torch.rand(shape, generator=gen).to(DEV)
        mom = torch.randn(shape, generator=gen).to(DEV)

        def run_update(g, expect_reuse):
            before = dict(_hip.stats)
            d, m = torch.zeros(shape, device=DEV), mom.clone()
            _hip.mi_update(g, m, m, d, data, 1.0, ALPHA, EPS)
            key = "partials_reused" if expect_reuse else "k1_passes"
            assert _hip.stats[key] == before[key] + 1
            return d, m

        def check(out, exact):
            _check_registered(out, exact)
            d1, m1 = run_update(out, True)                 # consumes the registered sums
            d2, m2 = run_update(out.clone(), False)        # another tensor: own K1 pass
            if exact:
                assert torch.equal(d1, d2) and torch.equal(m1, m2)
            else:
                assert_momentum_close(host(m1), host(m2), host(out), host(mom), 1.0)
                assert_delta_equal(host(d1), host(d2), host(m2))

        # TIM convolution
        grad = (torch.randn(shape, generator=gen) * 1e-3).to(DEV)
        w = torch.rand(15, 15, generator=gen)
        out = torch.empty(shape, device=DEV)
        _hip.depthwise_conv2d_same(grad, out, (w / w.sum()).to(DEV))
        check(out, False)
        # DIM backward (lane kernel at 1.1, table kernel at 2.0)
        size = shape[-1]
        if shape[-1] == shape[-2]:
            for resize, rnd, top, left in ((int(size * 1.1), size + 9, 3, 5), (2 * size, size + 30, 7, 1)):
                gx = torch.empty(shape, device=DEV)
                _hip.dim_bwd(grad, gx, resize, rnd, top, left)
                check(gx, False)
        # SIM / EMI / Admix backward, ensemble member sum: K1's own tiling
        gy = (torch.randn((5 * n,) + shape[1:], generator=gen) * 1e-3).to(DEV)
        gx = torch.empty(shape, device=DEV)
        _hip.scale_copies_bwd(gy, gx, 5)
        check(gx, True)
        _hip.sum_copies_bwd(gy, gx, 5)
        check(gx, True)
        gy = (torch.randn((6 * n,) + shape[1:], generator=gen) * 1e-3).to(DEV)
        _hip.admix_bwd(gy, gx, 3, 2)
        check(gx, True)
        members = [(torch.randn(shape, generator=gen) * 1e-3).to(DEV) for _ in range(4)]
        _hip.sum_members(members, gx)
        check(gx, True)
        # a kernel of the binding that overwrites the gradient drops the sums
        _hip.sum_members(members, gx)
        _hip.axpy(members[0], members[1], 0.5, gx)
        assert _hip.partials_of(gx) is None


def test_sum_members():
    """the ensemble's fan-out backward: the members' gradients added last member first, like autograd's input buffer"""
    gen = torch.Generator().manual_seed(6)
    for shape, m in (((3, 3, 224, 224), 4), ((2, 3, 7, 9), 3), ((1, 1, 1, 5), 2), ((2, 3, 31, 33), 8), ((2, 3, 31, 33), 9),
                     ((1, 3, 16, 20), 20)):           # more than eight members: folded eight at a time, same order
        gs = [torch.randn(shape, generator=gen) for _ in range(m)]
        ref = gs[-1].clone()
        for g in reversed(gs[:-1]):
            ref = ref + g
        gx = torch.empty(shape, device=DEV)
        _hip.sum_members([g.to(DEV) for g in gs], gx)
        assert np.array_equal(host(gx), ref.numpy())
    # and through autograd: x feeding several modules == torch's own accumulation
    from transferattack_amd.utils import _FanOut
    x = torch.randn(2, 3, 16, 16, generator=gen)
    ws = [torch.randn(2, 3, 16, 16, generator=gen) for _ in range(4)]
    r = torch.randn(2, 3, 16, 16, generator=gen)
    xin = x.clone().requires_grad_(True)
    ref = torch.autograd.grad((torch.stack([(xin * w).tanh() for w in ws]).mean(0) * r).sum(), xin)[0]
    xd = x.to(DEV).requires_grad_(True)
    views = _FanOut.apply(xd, 4)
    got = torch.autograd.grad((torch.stack([(v * w.to(DEV)).tanh() for v, w in zip(views, ws)]).mean(0) * r.to(DEV)).sum(), xd)[0]
    np.testing.assert_allclose(host(got), ref.numpy(), rtol=1e-5, atol=1e-7)     # tanh differs in the last bit across devices


def test_fused_update_under_graph_capture():
    shape = (8, 3, 224, 224)
    x, grad = torch.rand(shape, device=DEV), torch.randn(shape, device=DEV)
    d, m = torch.zeros(shape, device=DEV), torch.zeros(shape, device=DEV)
    d_ref, m_ref = d.clone(), m.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _hip.mi_update(grad, m, m, d, x, 1.0, ALPHA, EPS)          # warm-up on the side stream (allocates scratch)
    torch.cuda.current_stream().wait_stream(s)
    d.zero_(); m.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        _hip.mi_update(grad, m, m, d, x, 1.0, ALPHA, EPS)
    d.zero_(); m.zero_()
    for _ in range(3):
        graph.replay()
        _hip.mi_update(grad, m_ref, m_ref, d_ref, x, 1.0, ALPHA, EPS)
    torch.cuda.synchronize()
    assert torch.equal(d, d_ref) and torch.equal(m, m_ref)


@pytest.mark.parametrize("shape", [(8, 3, 224, 224), (3, 3, 37, 41), (2, 1, 5, 7)])
def test_normalize_and_producer_side_partials(shape):
    """PreprocessingModel's Normalize as HIP kernels: forward/backward bit-identical to the torch expression the
    reference evaluates ((x-mean)/std, grad/std), and the |g| tile sums the backward leaves behind make the fused
    update produce exactly what it produces when it runs its own K1 pass."""
    from transferattack_amd.utils import _Normalize
    gen = torch.Generator().manual_seed(shape[0])
    c = shape[1]
    mean, std = [0.485, 0.456, 0.406][:c], [0.229, 0.224, 0.225][:c]
    norm = _Normalize(mean, std).to(DEV)
    x = torch.rand(shape, generator=gen)
    gy = torch.randn(shape, generator=gen) * 1e-3
    xin = x.clone().requires_grad_(True)
    y_ref = (xin - torch.tensor(mean).view(1, -1, 1, 1)) / torch.tensor(std).view(1, -1, 1, 1)
    gx_ref = torch.autograd.grad(y_ref, xin, gy)[0]
    xd = x.to(DEV).requires_grad_(True)
    y = norm(xd)
    assert np.array_equal(host(y), y_ref.detach().numpy())
    gx = torch.autograd.grad(y, xd, gy.to(DEV))[0]
    assert np.array_equal(host(gx), gx_ref.numpy())
    assert _hip.partials_of(gx) is not None          # attached by the Normalize backward, survived autograd.grad
    mom, data = torch.randn(shape, generator=gen).to(DEV), torch.rand(shape, generator=gen).to(DEV)
    d1, m1 = torch.zeros(shape, device=DEV), mom.clone()
    _hip.mi_update(gx, m1, m1, d1, data, 1.0, ALPHA, EPS)                 # consumes the attached partials
    assert _hip.partials_of(gx) is None
    d2, m2 = torch.zeros(shape, device=DEV), mom.clone()
    _hip.mi_update(gx.clone(), m2, m2, d2, data, 1.0, ALPHA, EPS)         # different tensor -> own K1 pass
    assert torch.equal(d1, d2) and torch.equal(m1, m2)
    # an in-place edit of the gradient invalidates the registered sums
    gx2 = torch.autograd.grad(norm(xd), xd, gy.to(DEV))[0]
    gx2.mul_(2.0)
    d3, m3 = torch.zeros(shape, device=DEV), mom.clone()
    _hip.mi_update(gx2, m3, m3, d3, data, 1.0, ALPHA, EPS)
    d4, m4 = torch.zeros(shape, device=DEV), mom.clone()
    _hip.mi_update(gx2.clone(), m4, m4, d4, data, 1.0, ALPHA, EPS)
    assert torch.equal(d3, d4) and torch.equal(m3, m4)


def test_quantiser(golden):
    g = golden("update_stack")
    x, d = dev(g["x"]), dev(g["delta_d1"])
    out = torch.empty((3, 64, 64, 3), dtype=torch.uint8, device=DEV)
    _hip.quantize_u8_nhwc(x, d, out)
    assert np.array_equal(host(out), g["u8_d1"])
    gen = torch.Generator().manual_seed(3)
    for shape in ((4, 3, 224, 224), (2, 3, 5, 7), (2, 1, 9, 9)):
        x = torch.randint(0, 256, shape, generator=gen).float() / 255
        d = O.box_clamp((torch.rand(shape, generator=gen) - 0.5) * 2 * EPS, 0 - x, 1 - x)
        out = torch.empty((shape[0], shape[2], shape[3], shape[1]), dtype=torch.uint8, device=DEV)
        _hip.quantize_u8_nhwc(x.to(DEV), d.to(DEV), out)
        assert np.array_equal(host(out), O.quantize_u8(x + d))
        assert np.array_equal(host(out), C.quantize_u8_nhwc(x.numpy(), d.numpy()))


# --------------------------------------------------------------------------------------------------- TIM
def test_tim_golden(golden):
    g = golden("tim")
    out = torch.empty(g["grad_in"].shape, device=DEV)
    _hip.depthwise_conv2d_same(dev(g["grad_in"]), out, dev(g["kernel_gaussian"][0, 0]))
    assert np.array_equal(host(out), g["grad_out"])


@pytest.mark.parametrize("shape,k", [((4, 3, 224, 224), 15), ((2, 3, 299, 299), 15), ((2, 3, 37, 41), 15),
                                     ((2, 3, 64, 64), 3), ((2, 3, 64, 64), 5), ((2, 3, 64, 64), 7),
                                     ((2, 3, 50, 70), 9), ((1, 3, 33, 33), 4)])
def test_tim_random(shape, k):
    gen = torch.Generator().manual_seed(k)
    grad = torch.randn(shape, generator=gen)
    w = torch.rand(k, k, generator=gen)
    w = w / w.sum()
    out = torch.empty(shape, device=DEV)
    _hip.depthwise_conv2d_same(grad.to(DEV), out, w.to(DEV))
    assert np.array_equal(host(out), C.depthwise_conv2d_same(grad.numpy(), w.numpy()))     # FMA chain, bit-exact
    ref = torch.nn.functional.conv2d(grad, w[None, None].repeat(shape[1], 1, 1, 1), padding="same", groups=shape[1])
    np.testing.assert_allclose(host(out), ref.numpy(), rtol=0, atol=2e-6)


# --------------------------------------------------------------------------------------------------- DIM
def test_dim_golden(golden):
    g = golden("dim")
    x, gy = dev(g["x"]), dev(g["gy"])
    size = x.shape[-1]
    resize = int(size * float(g["resize_rate"]))
    for i, seed in enumerate(g["seeds"]):
        if g["identity"][i]:
            continue
        torch.manual_seed(int(seed))
        _, rnd, top, left = O.dim_draw(size, float(g["resize_rate"]), float(g["diversity_prob"]))
        y, gx = torch.empty_like(x), torch.empty_like(x)
        _hip.dim_fwd(x, y, resize, rnd, top, left)
        _hip.dim_bwd(gy, gx, resize, rnd, top, left)
        assert ulp_diff(host(y), g["y"][i]) <= 2
        assert np.array_equal(host(gx), g["gx"][i])


@pytest.mark.parametrize("size,rate,geoms", [
    (224, 1.1, [(224, 0, 0), (224, 22, 22), (245, 0, 1), (245, 1, 0), (237, 3, 5), (230, 16, 0)]),
    (64, 1.5, [(64, 0, 31), (95, 0, 0), (80, 7, 9)]),
    (33, 2.0, [(40, 5, 20), (65, 0, 1)]),
])
def test_dim_random(size, rate, geoms):
    gen = torch.Generator().manual_seed(size)
    resize = int(size * rate)
    x = torch.rand(3, 3, size, size, generator=gen)
    gy = torch.randn(3, 3, size, size, generator=gen)
    for rnd, top, left in geoms:
        geom = (True, rnd, top, left)
        y, gx = torch.empty(x.shape, device=DEV), torch.empty(x.shape, device=DEV)
        _hip.dim_fwd(x.to(DEV), y, resize, rnd, top, left)
        _hip.dim_bwd(gy.to(DEV), gx, resize, rnd, top, left)
        assert np.array_equal(host(y), C.dim_fwd(x.numpy(), geom, resize)), geom
        assert np.array_equal(host(gx), C.dim_bwd(gy.numpy(), geom, resize)), geom
        xin = x.clone().requires_grad_(True)
        yt = O.dim_apply(xin, geom, rate)
        assert ulp_diff(host(y), yt.detach().numpy()) <= (2 if size >= 128 else 4)   # ATen's own small-tensor variant
        np.testing.assert_allclose(host(gx), torch.autograd.grad(yt, xin, gy)[0].numpy(), rtol=0, atol=1e-6)


# -------------------------------------------------------------------------------------------- SIM / Admix
def test_sim_admix_golden(golden):
    g = golden("copies")
    x = dev(g["x"])
    y = torch.empty(g["sim_y"].shape, device=DEV)
    _hip.scale_copies_fwd(x, y, 5)
    assert np.array_equal(host(y), g["sim_y"])
    gx = torch.empty_like(x)
    _hip.scale_copies_bwd(dev(g["sim_gy"]), gx, 5)
    assert np.array_equal(host(gx), g["sim_gx"])
    torch.manual_seed(int(g["admix_seed"]))
    perm = torch.cat(O.admix_draw(x.shape[0])).to(DEV)
    y = torch.empty(g["admix_y"].shape, device=DEV)
    _hip.admix_fwd(x, perm, y, 3, 5, 0.2)
    assert np.array_equal(host(y), g["admix_y"])
    _hip.admix_bwd(dev(g["admix_gy"]), gx, 3, 5)
    assert np.array_equal(host(gx), g["admix_gx"])


def test_sum_copies_bwd():
    gen = torch.Generator().manual_seed(4)
    for shape, copies in (((3, 3, 224, 224), 11), ((2, 3, 7, 9), 5), ((1, 1, 1, 5), 2)):
        gy = torch.randn((copies * shape[0],) + shape[1:], generator=gen)
        x = torch.zeros(shape, requires_grad=True)
        ref = torch.autograd.grad(torch.cat([x + float(i) for i in range(copies)]), x, gy)[0]    # autograd's own order
        gx = torch.empty(shape, device=DEV)
        _hip.sum_copies_bwd(gy.to(DEV), gx, copies)
        assert np.array_equal(host(gx), ref.numpy())


def test_sim_admix_ragged():
    gen = torch.Generator().manual_seed(1)
    for shape in ((3, 3, 224, 224), (2, 3, 7, 9), (1, 1, 1, 5)):
        x = torch.rand(shape, generator=gen)
        perms = [torch.randperm(shape[0], generator=gen) for _ in range(2)]
        xin = x.clone().requires_grad_(True)
        y_ref = O.admix_copies(xin, perms, 0.3, 4)
        gy = torch.randn(y_ref.shape, generator=gen)
        gx_ref = torch.autograd.grad(y_ref, xin, gy)[0]
        y, gx = torch.empty(y_ref.shape, device=DEV), torch.empty(shape, device=DEV)
        _hip.admix_fwd(x.to(DEV), torch.cat(perms).to(DEV), y, 2, 4, 0.3)
        _hip.admix_bwd(gy.to(DEV), gx, 2, 4)
        assert np.array_equal(host(y), y_ref.detach().numpy()) and np.array_equal(host(gx), gx_ref.numpy())
        xin = x.clone().requires_grad_(True)
        y_ref = O.sim_copies(xin, 3)
        gy = torch.randn(y_ref.shape, generator=gen)
        y, gx = torch.empty(y_ref.shape, device=DEV), torch.empty(shape, device=DEV)
        _hip.scale_copies_fwd(x.to(DEV), y, 3)
        _hip.scale_copies_bwd(gy.to(DEV), gx, 3)
        assert np.array_equal(host(y), y_ref.detach().numpy())
        assert np.array_equal(host(gx), torch.autograd.grad(y_ref, xin, gy)[0].numpy())


# ------------------------------------------------------------------------------------------- VMI / NI / init
def test_vmi_kernels_and_philox():
    gen = torch.Generator().manual_seed(2)
    shape = (3, 3, 31, 33)                                              # numel not a multiple of 4
    x, d = torch.rand(shape, generator=gen), (torch.rand(shape, generator=gen) - 0.5) * EPS
    noise = (torch.rand(shape, generator=gen) - 0.5) * 3 * EPS
    out = torch.empty(shape, device=DEV)
    _hip.vmi_neighbor(x.to(DEV), d.to(DEV), out, 1.5 * EPS, noise=noise.to(DEV))
    assert np.array_equal(host(out), (x + d + noise).numpy())           # injected-noise mode == reference expression
    _hip.vmi_neighbor(x.to(DEV), d.to(DEV), out, 1.5 * EPS, seed=1234, offset=7)
    stream = C.philox_uniform(x.numel(), 1234, 7, np.float32(1.5 * EPS)).reshape(shape)
    assert np.array_equal(host(out), (x + d).numpy() + stream)          # in-kernel Philox == restated stream
    assert np.abs(stream).max() <= 1.5 * EPS and abs(float(stream.mean())) < 0.01 * EPS * 10
    big = C.philox_uniform(1 << 20, 5, 0, np.float32(1.0))
    assert abs(big.mean()) < 5e-3 and abs(big.std() - 1 / np.sqrt(3)) < 5e-3
    acc, g1, g2 = torch.empty(shape, device=DEV), torch.randn(shape, generator=gen), torch.randn(shape, generator=gen)
    _hip.grad_accumulate(acc, g1.to(DEV), first=True)
    _hip.grad_accumulate(acc, g2.to(DEV), first=False)
    assert np.array_equal(host(acc), (g1 + g2).numpy())
    var = torch.empty(shape, device=DEV)
    _hip.variance_finalize(acc, g1.to(DEV), var, 20)
    assert np.array_equal(host(var), ((g1 + g2) / 20 - g1).numpy())     # vmifgsm.py:58
    _hip.axpy(x.to(DEV), g1.to(DEV), ALPHA * 1.0, out)
    assert np.array_equal(host(out), (x + ALPHA * 1.0 * g1).numpy())    # nifgsm.py:39
    dl = torch.empty(shape, device=DEV)
    _hip.init_delta_uniform(dl, x.to(DEV), EPS, noise=noise.clamp(-EPS, EPS).to(DEV))
    assert np.array_equal(host(dl), O.delta_init(x, EPS, True, noise=noise.clamp(-EPS, EPS)).numpy())
    _hip.init_delta_uniform(dl, x.to(DEV), EPS, seed=9, offset=1)
    ref = O.box_clamp(torch.from_numpy(C.philox_uniform(x.numel(), 9, 1, np.float32(EPS)).reshape(shape)), 0 - x, 1 - x)
    assert np.array_equal(host(dl), ref.numpy())


def test_streaming_kernels_at_batch_size():
    """VMI / NI / random-start / accumulate kernels at the reference's batch shape (32 x 3 x 224 x 224), against the
    reference's torch expressions (bit-exact) and the restated Philox stream"""
    gen = torch.Generator().manual_seed(12)
    shape = (32, 3, 224, 224)
    x = torch.randint(0, 256, shape, generator=gen).float() / 255
    d = (torch.rand(shape, generator=gen) - 0.5) * EPS
    noise = (torch.rand(shape, generator=gen) - 0.5) * 3 * EPS
    g1, g2 = torch.randn(shape, generator=gen) * 1e-3, torch.randn(shape, generator=gen) * 1e-3
    xd, dd = x.to(DEV), d.to(DEV)
    out = torch.empty(shape, device=DEV)
    _hip.vmi_neighbor(xd, dd, out, 1.5 * EPS, noise=noise.to(DEV))
    assert np.array_equal(host(out), (x + d + noise).numpy())
    _hip.vmi_neighbor(xd, dd, out, 1.5 * EPS, seed=77, offset=3)
    stream = C.philox_uniform(x.numel(), 77, 3, np.float32(1.5 * EPS)).reshape(shape)
    assert np.array_equal(host(out), (x + d).numpy() + stream)
    acc = torch.empty(shape, device=DEV)
    _hip.grad_accumulate(acc, g1.to(DEV), first=True)
    _hip.grad_accumulate(acc, g2.to(DEV), first=False)
    assert np.array_equal(host(acc), (g1 + g2).numpy())
    var = torch.empty(shape, device=DEV)
    _hip.variance_finalize(acc, g1.to(DEV), var, 20)
    assert np.array_equal(host(var), ((g1 + g2) / 20 - g1).numpy())
    _hip.axpy(xd, g1.to(DEV), ALPHA, out)
    assert np.array_equal(host(out), (x + ALPHA * g1).numpy())
    _hip.init_delta_uniform(out, xd, EPS, noise=noise.clamp(-EPS, EPS).to(DEV))
    assert np.array_equal(host(out), O.delta_init(x, EPS, True, noise=noise.clamp(-EPS, EPS)).numpy())
    _hip.init_delta_uniform(out, xd, EPS, seed=9, offset=1)
    ref = O.box_clamp(torch.from_numpy(C.philox_uniform(x.numel(), 9, 1, np.float32(EPS)).reshape(shape)), 0 - x, 1 - x)
    assert np.array_equal(host(out), ref.numpy())
    # momentum with a variance term (VMI: get_momentum(grad + variance)) at this size
    m_prev = torch.randn(shape, generator=gen)
    m_out = torch.empty(shape, device=DEV)
    _hip.momentum(g1.to(DEV), m_prev.to(DEV), m_out, 1.0, variance=g2.to(DEV))
    assert_momentum_close(host(m_out), O.momentum_step(g1 + g2, m_prev, 1.0).numpy(), (g1 + g2).numpy(), m_prev.numpy(), 1.0)


def test_bad_arguments_fail_loudly():
    x = torch.rand(2, 3, 8, 8, device=DEV)
    with pytest.raises(_hip.HipExtensionError):
        _hip.depthwise_conv2d_same(x, x, torch.rand(3, 3, device=DEV))          # aliasing
    with pytest.raises(_hip.HipExtensionError):
        _hip.dim_fwd(x, torch.empty_like(x), 8, 9, 0, 0)                        # rnd > resize
    with pytest.raises(_hip.HipExtensionError):
        _hip.momentum(x.cpu(), None, torch.empty_like(x), 1.0)                  # CPU tensor
    with pytest.raises(TypeError):
        _hip.momentum(x.double(), None, torch.empty_like(x), 1.0)
