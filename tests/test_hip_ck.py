"""GPU (-m gpu): libta_ck.so (include/ta_ck.h) kernel by kernel, through the C-ABI -- every tile configuration of every epilogue
against the two-kernel expression it replaces, evaluated by torch on the device (MIOpen convolution + the glue arithmetic of
csrc/glue.hip, whose own bit-exactness tests/test_hip_kernels.py holds).  The convolution's accumulation order is the kernel's
own, as it is MIOpen's in the two-kernel form, so the bound is fp32 rounding of a K-term sum: 2e-5 of max|result|; the
epilogue's own arithmetic -- which operand is added first, where the clamp / threshold sits, NaN behaviour -- is checked exactly
on a problem whose accumulations are exact (small integers)."""
import pytest
import torch
import torch.nn.functional as F

from transferattack_amd import _ck

pytestmark = pytest.mark.gpu
DEV = "cuda"
CL = torch.channels_last


def nhwc(t):
    return t.to(DEV).contiguous(memory_format=CL)


def run_all(kind, geom, a, w, d0, d1, d2, out_shape):
    """every configuration that takes the problem -> list of (name, result)"""
    lib = _ck.load()
    got = []
    for idx in range(lib.ta_ck_instances(kind, geom[5], geom[6], geom[7])):
        e = torch.full(out_shape, float("nan"), device=DEV).contiguous(memory_format=CL)
        if _ck.conv(kind, idx, a, w, d0, d1, d2, e, geom, probing=True) == 0:
            got.append((lib.ta_ck_instance_name(kind, geom[5], geom[6], geom[7], idx).decode(), e))
    torch.cuda.synchronize()
    return got


@pytest.mark.parametrize("n,cin,cout,hw,ks,stride,pad", [(3, 64, 64, 14, 1, 1, 0), (2, 128, 32, 9, 3, 1, 1), (2, 64, 128, 12, 3, 2, 1),
                                                          (5, 256, 64, 7, 1, 1, 0), (2, 4, 64, 20, 7, 2, 3)])
def test_forward_epilogues(n, cin, cout, hw, ks, stride, pad):
    gen = torch.Generator().manual_seed(n * cin + cout + ks)
    x = nhwc(torch.randn(n, cin, hw, hw + 1, generator=gen))
    conv = torch.nn.Conv2d(cin, cout, ks, stride, pad).to(DEV)
    geom = _ck.geometry(x.shape, conv)
    w = _ck.weight_kyxc(conv)
    y = F.conv2d(x, conv.weight, None, stride, pad)
    other = nhwc(torch.randn(y.shape, generator=gen))
    bias2 = torch.randn(cout, generator=gen).to(DEV)
    b = conv.bias.view(1, -1, 1, 1)
    cases = [(_ck.FWD_BIAS_RELU, (conv.bias, None, None), torch.clamp_min(y + b, 0))]
    if ks == 1 and stride == 1:
        cases += [(_ck.FWD_BIAS_ADD_RELU, (conv.bias, other, None), torch.clamp_min((y + b) + other, 0)),
                  (_ck.FWD_BIAS_ADD_BIAS_RELU, (conv.bias, other, bias2), torch.clamp_min((y + b) + (other + bias2.view(1, -1, 1, 1)), 0))]
    for kind, (d0, d1, d2), want in cases:
        got = run_all(kind, geom, x, w, d0, d1, d2, tuple(y.shape))
        assert len(got) >= 3, "kind %d: only %d configurations take %s" % (kind, len(got), geom)
        scale = float(want.abs().max())
        for name, e in got:
            assert not torch.isnan(e).any(), name
            assert float((e - want).abs().max()) <= 2e-5 * scale, (kind, name)
            assert float(((e == 0) != (want == 0)).float().mean()) <= 1e-4, name            # the clamp sits where it should


@pytest.mark.parametrize("n,cin,cout,hw,ks,pad", [(3, 64, 256, 14, 1, 0), (2, 32, 64, 9, 3, 1), (4, 256, 64, 7, 1, 0), (2, 48, 48, 10, 5, 2)])
def test_input_gradient_epilogues(n, cin, cout, hw, ks, pad):
    """TA_CK_FWD_MASK / _ADD_MASK on the rewritten problem (_ck.backward_as_forward) == convolution_backward (input) + threshold"""
    gen = torch.Generator().manual_seed(n + cin + cout + ks)
    conv = torch.nn.Conv2d(cin, cout, ks, 1, pad, bias=False).to(DEV)
    act = nhwc(torch.randn(n, cin, hw, hw + 2, generator=gen))                 # the activation in front of conv (its sign is the mask)
    g = nhwc(torch.randn(n, cout, hw, hw + 2, generator=gen))
    other = nhwc(torch.randn(act.shape, generator=gen))
    geom = _ck.geometry(act.shape, conv)
    fgeom = _ck.backward_as_forward(geom)
    wt = _ck.weight_flipped_cyxk(conv)
    gx = torch.ops.aten.convolution_backward(g, act, conv.weight, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    for kind, (d0, d1), want in ((_ck.FWD_MASK, (act, None), torch.ops.aten.threshold_backward(gx, act, 0)),
                                 (_ck.FWD_ADD_MASK, (other, act), torch.ops.aten.threshold_backward(gx + other, act, 0))):
        got = run_all(kind, fgeom, g, wt, d0, d1, None, tuple(act.shape))
        assert len(got) >= 3
        scale = float(want.abs().max())
        for name, e in got:
            assert float((e - want).abs().max()) <= 2e-5 * scale, (kind, name)
            assert torch.equal(e == 0, want == 0) or float(((e == 0) != (want == 0)).float().mean()) <= 1e-5, name


def test_epilogue_arithmetic_is_exact_where_the_sums_are():
    """small-integer operands: every accumulation is exact in fp32, so ANY order gives the same sum and the results must EQUAL the
    two-kernel expression bit for bit -- rounding order of the epilogue ((acc + b) + o, not acc + (b + o)), clamp, threshold, and
    the NaN rules of clamp_min_ / threshold_backward (a NaN sum stays NaN; a NaN activation lets the gradient pass)"""
    gen = torch.Generator().manual_seed(5)
    n, cin, cout, hw = 2, 64, 64, 8
    ints = lambda *shape: torch.randint(-3, 4, shape, generator=gen).float()      # noqa: E731
    x, other = nhwc(ints(n, cin, hw, hw)), nhwc(ints(n, cout, hw, hw) + 0.25)
    conv = torch.nn.Conv2d(cin, cout, 1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(ints(cout, cin, 1, 1))
        conv.bias.copy_(ints(cout) * 1e-3 + 1e8)                                # (acc + b) + o  !=  acc + (b + o) in fp32 for these
    other = other - 1e8
    geom, w = _ck.geometry(x.shape, conv), _ck.weight_kyxc(conv)
    y = F.conv2d(x, conv.weight, None)
    want = torch.clamp_min((y + conv.bias.view(1, -1, 1, 1)) + other, 0)
    wrong = torch.clamp_min(y + (conv.bias.view(1, -1, 1, 1) + other), 0)
    assert not torch.equal(want, wrong), "the fixture does not tell the two orders apart"
    for name, e in run_all(_ck.FWD_BIAS_ADD_RELU, geom, x, w, conv.bias, other, None, tuple(y.shape)):
        assert torch.equal(e, want), name
    # NaN rules
    with torch.no_grad():
        conv.bias.copy_(ints(cout))
    xn = x.clone()
    xn[0, :, 0, 0] = float("nan")
    yn = torch.clamp_min(F.conv2d(xn, conv.weight, None) + conv.bias.view(1, -1, 1, 1), 0)
    for name, e in run_all(_ck.FWD_BIAS_RELU, geom, xn, w, conv.bias, None, None, tuple(y.shape)):
        assert torch.equal(torch.isnan(e), torch.isnan(yn)) and bool(torch.isnan(e[0, :, 0, 0]).all()), name
        assert torch.equal(torch.nan_to_num(e), torch.nan_to_num(yn)), name
    act = nhwc(ints(n, cin, hw, hw))
    act[1, :, 1, 1] = float("nan")
    g = nhwc(ints(n, cout, hw, hw))
    fgeom, wt = _ck.backward_as_forward(geom), _ck.weight_flipped_cyxk(conv)
    gx = torch.ops.aten.convolution_backward(g, act, conv.weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    want = torch.ops.aten.threshold_backward(gx, act, 0)
    for name, e in run_all(_ck.FWD_MASK, fgeom, g, wt, act, None, None, tuple(act.shape)):
        assert torch.equal(e, want), name                                       # incl. the NaN activation: the gradient passes


def test_site_decisions_persist(tmp_path, monkeypatch):
    """the tuner's decision for a site is written to the plan file and taken from there by a process that has not tuned (here:
    the same process with its in-memory plans dropped) -- as MIOpen's find results persist in its user find-db"""
    import json
    from transferattack_amd.backbones import fused
    monkeypatch.setenv("TA_CK_EPILOGUE", "1")
    monkeypatch.setenv("TA_CK_PLAN_CACHE", str(tmp_path / "plans.json"))
    monkeypatch.setattr(_ck, "_disk", None)
    monkeypatch.setattr(_ck, "plans", {})
    conv = torch.nn.Conv2d(64, 64, 1).to(DEV)
    x = nhwc(torch.randn(8, 64, 28, 28))
    want = torch.clamp_min(F.conv2d(x, conv.weight, conv.bias), 0)
    before = dict(_ck.stats)
    y1, _ = fused._site_bias_relu(x, conv, lambda t: None)
    assert _ck.stats["tuned_sites"] == before["tuned_sites"] + 1
    stored = json.load(open(tmp_path / "plans.json"))
    assert len(stored) == 1
    monkeypatch.setattr(_ck, "_disk", None)
    monkeypatch.setattr(_ck, "plans", {})
    conv.__dict__.pop("_ta_ck_sites")
    y2, _ = fused._site_bias_relu(x, conv, lambda t: None)
    assert _ck.stats["tuned_sites"] == before["tuned_sites"] + 1 and _ck.stats["sites_from_disk"] == before["sites_from_disk"] + 1
    for y in (y1, y2):
        assert float((y - want).abs().max()) <= 2e-5 * float(want.abs().max())
