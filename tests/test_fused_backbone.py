"""The fused execution of a ResNet surrogate (backbones/fused.py + csrc/glue.hip) computes the module path's bits: same
convolutions, same rounding points, fewer passes.  CPU tier: the glue kernels from their host build (tests/host_kernels.py),
ATen's CPU convolutions on both sides -> logits and input gradient must be EQUAL.  The GPU tier repeats it on MI355X
(tests/test_hip_configs.py::test_fused_glue_is_the_same_surrogate)."""
import numpy as np
import pytest
import torch

import host_kernels
from transferattack_amd import _hip, backbones
from transferattack_amd.backbones import fused


def bias_as_on_rocm(net):
    """ATen's CPU convolution (oneDNN) adds the bias inside its accumulation; on ROCm the convolution is MIOpen's and the
    bias a separate fp32 add (`output.add_(bias)`, the elementwise kernel after every convolution in
    profiles/r03/steady_state_b125_r3a.json).  The module side of this comparison gets the ROCm arithmetic."""
    import types
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
            def forward(self, x):
                y = torch.nn.functional.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
                return y.add_(self.bias.view(1, -1, 1, 1))
            m.forward = types.MethodType(forward, m)


def both_paths(monkeypatch, name, x, label, channels_last):
    net = backbones.create(name, seed=0, verbose=False)
    for p in net.parameters():
        p.requires_grad_(False)
    backbones.fold_batchnorm(net)
    bias_as_on_rocm(net)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    out = {}
    for tag, flag in (("module", "0"), ("fused", "1")):
        monkeypatch.setenv("TA_FUSED_GLUE", flag)
        xin = x.clone().requires_grad_(True)
        logits = net(xin)
        loss = torch.nn.functional.cross_entropy(logits, label)
        out[tag] = (logits.detach().clone(), torch.autograd.grad(loss, xin)[0].contiguous().clone())
    return net, out


@pytest.mark.parametrize("name,channels_last", [("resnet18", False), ("resnet18", True), ("resnet50", True), ("resnet50", False)])
def test_fused_glue_equals_module_path(monkeypatch, name, channels_last):
    host_kernels.install(monkeypatch)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 64, 64, generator=gen)
    label = torch.randint(0, 1000, (2,), generator=gen)
    net, out = both_paths(monkeypatch, name, x, label, channels_last)
    assert getattr(net, "_bn_folded", False)
    assert torch.equal(out["module"][0], out["fused"][0]), "logits differ"
    if channels_last:          # the stem's input gradient comes from csrc/stem.hip there (its own accumulation order)
        g_m, g_f = out["module"][1].double(), out["fused"][1].double()
        assert float((g_m - g_f).norm() / g_m.norm()) <= 1e-6
        monkeypatch.setenv("TA_STEM_KERNEL", "0")
        net2, out2 = both_paths(monkeypatch, name, x, label, channels_last)
        assert torch.equal(out2["module"][1], out2["fused"][1]), "input gradient differs with MIOpen's stem backward"
    else:
        assert torch.equal(out["module"][1], out["fused"][1]), "input gradient differs"
    assert float(out["fused"][1].abs().max()) > 0


@pytest.mark.parametrize("n,oh,ow", [(2, 16, 16), (1, 9, 37), (1, 4, 33)])
def test_stem_input_grad_kernel(monkeypatch, n, oh, ow):
    """csrc/stem.hip (fp32 MFMA, four stride-2 phases in one GEMM) against ATen's convolution backward: same function, its own
    fixed accumulation order -> equal to fp32 rounding of a 3136-term sum, checked against an fp64 evaluation"""
    host_kernels.install(monkeypatch)
    gen = torch.Generator().manual_seed(7)
    w = torch.randn(64, 3, 7, 7, generator=gen) * 0.05
    dy = torch.randn(n, 64, oh, ow, generator=gen).contiguous(memory_format=torch.channels_last)
    x_like = torch.empty(n, 3, 2 * oh, 2 * ow)
    truth = torch.ops.aten.convolution_backward(dy.double().contiguous(), x_like.double(), w.double(), None, [2, 2], [3, 3], [1, 1],
                                                False, [0, 0], 1, [True, False, False])[0]
    ref32 = torch.ops.aten.convolution_backward(dy, x_like, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    got = _hip.stem7s2_input_grad(dy, _hip.stem7s2_prepare(w), torch.full_like(x_like, float("nan")))
    scale = float(truth.abs().max())
    err_got, err_ref = float((got.double() - truth).abs().max()) / scale, float((ref32.double() - truth).abs().max()) / scale
    assert not torch.isnan(got).any()
    assert err_got <= max(4 * err_ref, 2e-6), (err_got, err_ref)
    # dy in NCHW memory (the plain module path of an NCHW surrogate): other staging loads, same schedule -> the SAME bits, also
    # with the |dx / std| sums; and through the module path's autograd node
    std = torch.tensor([0.229, 0.224, 0.225])
    got_nchw = _hip.stem7s2_input_grad(dy.contiguous(), _hip.stem7s2_prepare(w), torch.full_like(x_like, float("nan")), std=std)
    assert torch.equal(got_nchw, got)
    got_cl = _hip.stem7s2_input_grad(dy, _hip.stem7s2_prepare(w), torch.full_like(x_like, float("nan")), std=std)
    sums_a, sums_b = _hip.partials_of(got_nchw), _hip.partials_of(got_cl)
    assert sums_a[1] == sums_b[1] and torch.equal(sums_a[0], sums_b[0])
    from transferattack_amd.backbones import fused, resnet
    net = resnet.resnet18().eval()
    for p_ in net.parameters():
        p_.requires_grad_(False)
    with torch.no_grad():
        net.conv1.weight.copy_(w)
    x = torch.randn(n, 3, 2 * oh, 2 * ow, generator=gen).requires_grad_(True)
    y = fused._StemConvFn.apply(x, net, None)
    assert torch.equal(y, net.conv1(x))
    assert torch.equal(torch.autograd.grad(y, x, dy.contiguous())[0], got)


@pytest.mark.parametrize("shape,k,s,p", [((2, 8, 16, 16), 3, 2, 1), ((1, 4, 9, 13), 3, 2, 1), ((2, 8, 12, 12), 2, 2, 0), ((1, 4, 11, 7), 3, 1, 1)])
def test_maxpool_backward_relu_kernel(monkeypatch, shape, k, s, p):
    """ta_maxpool_bwd_relu (junction add + max-pool backward + ReLU threshold as one gather) against the ATen passes it replaces"""
    host_kernels.install(monkeypatch)
    gen = torch.Generator().manual_seed(11)
    cl = torch.channels_last
    pre = torch.randn(shape, generator=gen)
    y = pre.clamp_min(0).contiguous(memory_format=cl)                       # post-ReLU map: plenty of exact zeros and ties at 0
    pooled, idx = torch.nn.functional.max_pool2d(y, k, s, p, return_indices=True)
    ga = torch.randn(pooled.shape, generator=gen).contiguous(memory_format=cl)
    gb = torch.randn(pooled.shape, generator=gen).contiguous(memory_format=cl)
    for second in (gb, None):
        g = ga if second is None else ga + second
        ref = torch.ops.aten.max_pool2d_with_indices_backward(g, y, [k, k], [s, s], [p, p], [1, 1], False, idx)
        ref = torch.ops.aten.threshold_backward(ref, y, 0)
        got = _hip.maxpool_bwd_relu(ga, idx.contiguous(memory_format=cl), y, torch.full_like(y, float("nan")), k, s, p, gb=second)
        assert not torch.isnan(got).any()
        assert float((got - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 16, 2, 6), (3, 64, 12, 10), (1, 8, 2, 2)])
def test_stem_pool_pair(monkeypatch, shape):
    """ta_maxpool3s2_fwd / ta_maxpool3s2_bwd_relu (the stem pool with a byte of argmax and the activation's pass bits) against
    max_pool2d_with_indices, its backward and threshold_backward: pooled values, the element that wins each window (ties at the
    ReLU's zeros and between equal positives included), the bits, and the gradient -- all EQUAL"""
    host_kernels.install(monkeypatch)
    gen = torch.Generator().manual_seed(sum(shape))
    cl = torch.channels_last
    n, c, h, w = shape
    pre = torch.randn(shape, generator=gen)
    y = (pre.clamp_min(0) * 4).round().div(4).contiguous(memory_format=cl)       # quarter steps: ties between positives too
    y[0, :, 0, 0] = float("nan")                                                  # a NaN wins its windows, as in ATen
    y[-1, 1, -1, -1] = float("inf")
    assert _hip.maxpool3s2_takes(y, torch.nn.MaxPool2d(3, 2, 1))
    assert not _hip.maxpool3s2_takes(y, torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True))
    assert not _hip.maxpool3s2_takes(y.contiguous(), torch.nn.MaxPool2d(3, 2, 1))
    want, idx = torch.nn.functional.max_pool2d(y, 3, 2, 1, return_indices=True)
    pooled, arg, bits = _hip.maxpool3s2_fwd(y)
    assert pooled.is_contiguous(memory_format=cl) and arg.is_contiguous(memory_format=cl)
    assert torch.equal(torch.nan_to_num(pooled, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
    kh, kw = arg.long() // 3, arg.long() % 3
    ii = torch.arange(h // 2).view(1, 1, -1, 1)
    jj = torch.arange(w // 2).view(1, 1, 1, -1)
    assert torch.equal((2 * ii - 1 + kh) * w + (2 * jj - 1 + kw), idx)
    passes = ~(y <= 0)                                                            # threshold_backward's test; NaN passes
    flat = passes.permute(0, 2, 3, 1).reshape(-1, 8).to(torch.uint8)
    assert torch.equal(bits, (flat << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8))
    ga = torch.randn(want.shape, generator=gen).contiguous(memory_format=cl)
    gb = torch.randn(want.shape, generator=gen).contiguous(memory_format=cl)
    for second in (gb, None):
        g = ga if second is None else ga + second
        ref = torch.ops.aten.max_pool2d_with_indices_backward(g, y, [3, 3], [2, 2], [1, 1], [1, 1], False, idx)
        ref = torch.ops.aten.threshold_backward(ref, y, 0)
        generic = _hip.maxpool_bwd_relu(ga, idx.contiguous(memory_format=cl), y, torch.full_like(y, float("nan")), 3, 2, 1, gb=second)
        got = _hip.maxpool3s2_bwd_relu(ga, arg, bits, torch.full_like(y, float("nan")), gb=second)
        assert not torch.isnan(got).any()
        assert torch.equal(got, generic)
        assert float((got - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        _hip.maxpool3s2_bwd_relu(ga, arg, bits[:-1], torch.empty_like(y))


def test_fused_path_steps_aside(monkeypatch):
    """hooks anywhere in the backbone, an unfolded BatchNorm, training mode or a trainable weight: the module path runs"""
    host_kernels.install(monkeypatch)
    monkeypatch.setenv("TA_FUSED_GLUE", "1")
    x = torch.rand(1, 3, 64, 64).requires_grad_(True)
    plain = backbones.create("resnet18", seed=0, verbose=False)
    assert not fused.usable(plain, x)                                  # BatchNorm not folded
    net = backbones.create("resnet18", seed=0, verbose=False)
    for p in net.parameters():
        p.requires_grad_(False)
    backbones.fold_batchnorm(net)
    assert fused.usable(net, x)
    calls = []
    handle = net.layer2[0].conv1.register_forward_hook(lambda m, i, o: calls.append(1))
    assert not fused.usable(net, x)
    net(x)
    assert calls == [1]                                                # the hook saw its call: module path
    handle.remove()
    assert fused.usable(net, x)
    net.fc.weight.requires_grad_(True)
    assert not fused.usable(net, x)
    net.fc.weight.requires_grad_(False)
    # sub-modules swapped, not hooked: the modules' own path must run
    for swap in (lambda n: setattr(n.layer1[0], "relu", torch.nn.LeakyReLU(0.1)),
                 lambda n: setattr(n, "maxpool", torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True)),
                 lambda n: setattr(n, "avgpool", torch.nn.AdaptiveMaxPool2d(1)),
                 lambda n: setattr(n.layer2[0], "downsample", torch.nn.Sequential(*n.layer2[0].downsample, torch.nn.Dropout(0.0))),
                 lambda n: setattr(n.layer3[1], "conv2", torch.nn.Sequential(n.layer3[1].conv2))):
        other = backbones.create("resnet18", seed=0, verbose=False)
        for p in other.parameters():
            p.requires_grad_(False)
        backbones.fold_batchnorm(other)
        assert fused.usable(other, x)
        swap(other)
        assert not fused.usable(other, x)
        other(x)
    monkeypatch.setenv("TA_FUSED_GLUE", "0")
    assert not fused.usable(net, x)


@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_fused_path_backpropagates_twice(monkeypatch, name):
    """two gradients through ONE forward with retain_graph (vaifgsm.py:49: one per auxiliary loss; adaea.py:44,51: per-member
    gradients, then the fused-logit gradient): the saved maps live in save_for_backward, not in ctx attributes cleared by the
    first backward; and without retain_graph the second backward is autograd's own error, not a TypeError"""
    host_kernels.install(monkeypatch)
    monkeypatch.setenv("TA_FUSED_GLUE", "1")
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 64, 64, generator=gen)
    label = torch.randint(0, 1000, (2,), generator=gen)
    net = backbones.create(name, seed=0, verbose=False)
    for p in net.parameters():
        p.requires_grad_(False)
    backbones.fold_batchnorm(net)
    bias_as_on_rocm(net)
    xin = x.clone().requires_grad_(True)
    assert fused.usable(net, xin)
    logits = net(xin)
    loss_a = torch.nn.functional.cross_entropy(logits, label)
    loss_b = logits.logsumexp(1).sum()
    g_a = torch.autograd.grad(loss_a, xin, retain_graph=True)[0].clone()
    g_b = torch.autograd.grad(loss_b, xin, retain_graph=True)[0].clone()
    g_a2 = torch.autograd.grad(loss_a, xin)[0]
    assert torch.equal(g_a, g_a2) and not torch.equal(g_a, g_b)
    monkeypatch.setenv("TA_FUSED_GLUE", "0")
    xm = x.clone().requires_grad_(True)
    g_bm = torch.autograd.grad(net(xm).logsumexp(1).sum(), xm)[0]
    assert torch.equal(g_b, g_bm)
    with pytest.raises(RuntimeError):
        torch.autograd.grad(loss_a, xin)                               # graph freed: autograd says so


@pytest.mark.parametrize("attack,kw", [("vaifgsm", dict(epoch=2)), ("mifgsm", dict(epoch=2))])
def test_retaining_attacks_run_in_the_bench_arrangement(monkeypatch, attack, kw):
    """TA_FOLD_BN=1 (bench.py's arrangement) with an attack that backpropagates more than once per forward"""
    import transferattack_amd as ta
    host_kernels.install(monkeypatch)
    monkeypatch.setenv("TA_FOLD_BN", "1")
    monkeypatch.setenv("TA_FUSED_GLUE", "1")
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 64, 64, generator=gen)
    label = torch.randint(0, 1000, (2,), generator=gen)
    atk = ta.load_attack_class(attack)(model_name="resnet18", **kw)
    assert getattr(atk.model[1], "_bn_folded", False)
    delta = atk(x, label)
    assert float(delta.abs().max()) <= 16 / 255 + 1e-7 and float(delta.abs().max()) > 0


def test_glue_kernels_against_aten(monkeypatch):
    host_kernels.install(monkeypatch)
    gen = torch.Generator().manual_seed(5)
    for fmt, hw in ((torch.contiguous_format, (6, 10)), (torch.channels_last, (6, 10)), (torch.contiguous_format, (7, 7)),
                    (torch.channels_last, (7, 7))):
        y = torch.randn(4, 8, *hw, generator=gen).contiguous(memory_format=fmt)
        o = torch.randn(4, 8, *hw, generator=gen).contiguous(memory_format=fmt)
        b, bo = torch.randn(8, generator=gen), torch.randn(8, generator=gen)
        y[0, 0, 0, 0] = float("nan")
        ref = (y + b.view(1, -1, 1, 1)).clamp_min(0)
        got = _hip.bias_act_(y.clone(memory_format=torch.preserve_format), b)
        assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True)
        got = _hip.bias_act_(y.clone(memory_format=torch.preserve_format), b, relu=False)
        assert np.array_equal(got.numpy(), (y + b.view(1, -1, 1, 1)).numpy(), equal_nan=True)
        ref = ((y + b.view(1, -1, 1, 1)) + (o + bo.view(1, -1, 1, 1))).clamp_min(0)
        got = _hip.bias_add_relu_(y.clone(memory_format=torch.preserve_format), b, o, bo)
        assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True)
        ref = ((y + b.view(1, -1, 1, 1)) + o).clamp_min(0)
        got = _hip.bias_add_relu_(y.clone(memory_format=torch.preserve_format), b, o)
        assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True)
        ga, gb = torch.randn_like(y), torch.randn_like(y)
        res = y.clamp_min(0)
        ref = torch.ops.aten.threshold_backward(ga + gb, res, 0)
        got = _hip.relu_mask(ga, res, torch.empty_like(ga), gb=gb)
        assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True)
        ref = torch.ops.aten.threshold_backward(ga, res, 0)
        got = _hip.relu_mask(ga.clone(memory_format=torch.preserve_format), res, ga)
        assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True)
        # pass bits: the forward kernels leave one bit per element, the backward takes it instead of the activation
        for fwd in ("bias_act", "bias_add_relu"):
            yy = y.clone(memory_format=torch.preserve_format)
            bits = _hip.pass_bits_like(yy)
            bits.fill_(0xA5)
            res = _hip.bias_act_(yy, b, mask=bits) if fwd == "bias_act" else _hip.bias_add_relu_(yy, b, o, bo, mask=bits)
            flat = res.detach().as_strided((res.numel(),), (1,))                       # memory order
            want = np.packbits((~(flat <= 0)).numpy().astype(np.uint8), bitorder="little")
            assert np.array_equal(bits.numpy(), want), "pass bits differ from !(y <= 0)"
            for second in (None, gb):
                ref = torch.ops.aten.threshold_backward(ga if second is None else ga + second, res, 0)
                got = _hip.relu_mask(ga, res, torch.empty_like(ga), gb=second, mask=bits)
                assert np.array_equal(got.numpy(), ref.numpy(), equal_nan=True)
    with pytest.raises(_hip.HipExtensionError, match="numel"):                      # 4 * 3 * 3 * 1 = 36 elements: not a multiple of 8
        odd = torch.randn(4, 3, 3, 1)
        _hip.bias_act_(odd, torch.randn(3), mask=torch.empty(5, dtype=torch.uint8))


def test_attack_loops_through_the_fused_surrogate(monkeypatch):
    """whole loops (MI-FGSM; DTS = DIM o SIM on the input + TIM on the gradient; ensemble of a fused and an unfused member) with the fused
    ResNet execution against the same loops through the module path: identical perturbations on the CPU tier (with MIOpen's
    stem backward; the MFMA stem kernel has its own summation order)"""
    import transferattack_amd as ta
    host_kernels.install(monkeypatch)
    monkeypatch.setenv("TA_FOLD_BN", "1")
    monkeypatch.setenv("TA_CHANNELS_LAST", "1")
    monkeypatch.setenv("TA_STEM_KERNEL", "0")
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 64, generator=gen)
    label = torch.randint(0, 1000, (2,), generator=gen)
    for name, model, kw in (("mifgsm", "resnet18", dict(epoch=3)), ("dts", "resnet18", dict(epoch=2, diversity_prob=1.0)),
                            ("ens", ["resnet18", "mobilenet_v2"], dict(epoch=2))):
        out = {}
        for tag, flag in (("module", "0"), ("fused", "1")):
            monkeypatch.setenv("TA_FUSED_GLUE", flag)
            torch.manual_seed(123)
            atk = ta.load_attack_class(name)(model_name=model, **kw)
            for net in (atk.model.models if hasattr(atk.model, "models") else [atk.model]):
                bias_as_on_rocm(net)
            torch.manual_seed(456)
            out[tag] = atk(x, label)
        assert torch.equal(out["module"], out["fused"]), name
