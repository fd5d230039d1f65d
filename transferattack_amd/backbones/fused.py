"""Fused execution of a ResNet surrogate's forward + input-gradient backward (round 3).

What the attack loop needs from the surrogate is one thing, 10 to 210 times per batch: logits and d(loss)/d(input)
(transferattack/attack.py:104-122), weights frozen.  Through ``nn.Module`` + autograd every convolution of a ResNet with
folded BatchNorm is followed by a bias-add pass and a ReLU pass, every block by a residual-add pass, and the backward adds a
threshold pass per ReLU and an add per junction: 30 % of an iteration's GPU time at batch 125 is such memory-bound glue
(profiles/r03/steady_state_b125_r3a.json).  ``FusedResNet`` runs the SAME convolutions (``F.conv2d`` /
``aten::convolution_backward`` -> MIOpen, same algorithms) with the glue in fused HIP passes (csrc/glue.hip):

    forward    conv -> [bias + ReLU]                      1 pass instead of 2
               conv3, shortcut -> [bias + (bias) + add + ReLU]   1 pass instead of 3 (4 with a projection shortcut)
    backward   [junction add + threshold]                 1 pass instead of 2; thresholds in place on the convolution's output

as ONE ``autograd.Function`` over the whole network with a hand-written backward (only the input gradient exists: weights do
not require grad).  Every rounding point of the module path is kept, so logits and input gradient carry the module path's
bits (``tests/test_fused_backbone.py``: equality on the host stand-in and on the GPU).

It is an execution strategy of the surrogate, not a different surrogate: ``ResNet.forward`` takes it only when the module
tree is observed by nobody (no forward / backward hooks anywhere in the backbone -- model-related attacks that hook
``self.model[1]`` sub-modules get the plain module path), BatchNorm has been folded, and ``TA_FUSED_GLUE`` is not ``0``.

Round 6, ``TA_CK_EPILOGUE=1``: a convolution and the glue pass behind it become ONE composable_kernel convolution with the pass as
epilogue (libta_ck.so, ``_ck.py``) at every site where that measures faster than the two-kernel form -- forward (bias + ReLU, bias +
shortcut + ReLU, the stem on a 4-channel padded image) and backward (the forward kernel on the rewritten problem + threshold /
junction add).  And the plain MODULE path (``module_path_stem``) sends its stem's input gradient through csrc/stem.hip too.
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from .. import _ck, _hip


def _conv(x, conv):
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def _conv_input_grad(g, x_like, conv):
    """d/d(input) of ``conv`` for output gradient ``g``; ``x_like`` is the convolution's input (shape / memory format only)"""
    return torch.ops.aten.convolution_backward(g, x_like, conv.weight, None, list(conv.stride), list(conv.padding),
                                               list(conv.dilation), False, [0, 0], conv.groups, [True, False, False])[0]


# ---- convolution sites.  Each is "MIOpen convolution + one glue kernel" (the two-kernel form), or -- with TA_CK_EPILOGUE=1, NHWC
# operands and a shape for which it measured faster (``_ck.choose``) -- ONE composable_kernel convolution with the glue as its
# epilogue (libta_ck.so, include/ta_ck.h).  Same rounding points either way; the fused form leaves no pass bits (CK's epilogue
# operands are element tensors), so a backward site behind it reads the activation itself.  The decision and everything that
# only depends on (convolution, shapes) is kept on the convolution module: a decided site costs one dictionary lookup, one
# allocation and one ctypes call on the host (a ResNet-50 iteration has ~100 of them).
_CL = torch.channels_last


def _dense_nhwc(*tensors):
    return all(t is None or (t.is_contiguous(memory_format=_CL) and t.dtype == torch.float32 and t.is_cuda) for t in tensors)


def _site(conv, tag, x_shape):
    """the cached plan of this convolution for this role and input shape -- a tuple the site unpacks, or None where the two-kernel
    form runs -- or "new" when the site has not been decided yet"""
    cache = conv.__dict__.get("_ta_ck_sites")
    if cache is None:
        cache = conv.__dict__["_ta_ck_sites"] = {}
    key = (tag, tuple(x_shape), conv.weight._version)
    return cache.get(key, "new"), cache, key


def _site_bias_relu(x, conv, new_bits):
    """-> (clamp_min(conv(x) + bias, 0), pass bits | None)"""
    def two_kernels():
        y = _conv(x, conv)
        m = new_bits(y)
        _hip.bias_act_(y, conv.bias, mask=m)
        return y, m
    if not (_ck.enabled() and _dense_nhwc(x)):
        return two_kernels()
    plan, cache, key = _site(conv, "bias_relu", x.shape)
    if plan == "new":
        geom, plan = _ck.geometry(x.shape, conv), None
        if geom is not None:
            w, (ho, wo) = _ck.weight_kyxc(conv), _ck.out_hw(geom)
            y = torch.empty((geom[0], geom[4], ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
            run = lambda idx: _ck.conv(_ck.FWD_BIAS_RELU, idx, x, w, conv.bias, None, None, y, geom, probing=True)      # noqa: E731
            best = _ck.choose(("bias_relu", geom), [(_ck.FWD_BIAS_RELU, geom, run)], two_kernels)
            plan = None if best is None else (best[1], geom, w, ho, wo)
        cache[key] = plan
    if plan is None:
        return two_kernels()
    idx, geom, w, ho, wo = plan
    y = torch.empty((geom[0], geom[4], ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
    _ck.conv(_ck.FWD_BIAS_RELU, idx, x, w, conv.bias, None, None, y, geom)
    return y, None


def _site_stem_bias_relu(x, conv):
    """-> clamp_min(conv(x) + bias, 0) for the stem (a few input channels: 3).  composable_kernel's vector loads run along the
    channels, so its form of this site pads the image to 4 channels (one torch pass, 75 -> 100 MB at batch 125) and the filter
    with an all-zero fourth input plane: the products with it are exact zeros, the sum is unchanged.  Against MIOpen's
    zero fill + 3-channel convolution + ta_bias_act: 568 -> 377 + ~50 us at batch 125 (profiles/r06/ck_stem_forward_probe_r6k.txt)."""
    def two_kernels():
        return _hip.bias_act_(_conv(x, conv), conv.bias)
    cin = conv.weight.shape[1]
    if not (_ck.enabled() and _dense_nhwc(x)) or cin % 4 == 0 or cin > 3:
        return two_kernels()
    plan, cache, key = _site(conv, "stem_bias_relu", x.shape)

    def padded():
        return torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 4 - cin))            # [n, h, w, 4], dense

    if plan == "new":
        geom3, plan = _ck.geometry(x.shape, conv), None
        if geom3 is not None:
            geom = (geom3[0], 4) + geom3[2:]
            w4 = torch.nn.functional.pad(_ck.weight_kyxc(conv), (0, 4 - cin)).contiguous()
            ho, wo = _ck.out_hw(geom)
            y = torch.empty((geom[0], geom[4], ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)

            def run_with_pad(idx):            # what will run: the padding pass + the convolution
                return _ck.conv(_ck.FWD_BIAS_RELU, idx, padded(), w4, conv.bias, None, None, y, geom, probing=True)
            best = _ck.choose(("stem_bias_relu", geom), [(_ck.FWD_BIAS_RELU, geom, run_with_pad)], two_kernels)
            plan = None if best is None else (best[1], geom, w4, ho, wo)
        cache[key] = plan
    if plan is None:
        return two_kernels()
    idx, geom, w4, ho, wo = plan
    y = torch.empty((geom[0], geom[4], ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
    _ck.conv(_ck.FWD_BIAS_RELU, idx, padded(), w4, conv.bias, None, None, y, geom)
    return y


def _site_bias_add_relu(x, conv, other, bias_other, new_bits):
    """-> (clamp_min((conv(x) + bias) + (other [+ bias_other]), 0), pass bits | None): a block's last convolution, its shortcut, its ReLU"""
    def two_kernels():
        y = _conv(x, conv)
        m = new_bits(y)
        _hip.bias_add_relu_(y, conv.bias, other, bias_other, mask=m)
        return y, m
    if not (_ck.enabled() and _dense_nhwc(x, other)):
        return two_kernels()
    kind = _ck.FWD_BIAS_ADD_RELU if bias_other is None else _ck.FWD_BIAS_ADD_BIAS_RELU
    tag = "bias_add_relu" if bias_other is None else "bias_add_bias_relu"
    plan, cache, key = _site(conv, tag, x.shape)
    if plan == "new":
        geom, plan = _ck.geometry(x.shape, conv), None
        if geom is not None and tuple(other.shape) == (geom[0], geom[4]) + _ck.out_hw(geom):
            w = _ck.weight_kyxc(conv)
            y = torch.empty_like(other)
            run = lambda idx: _ck.conv(kind, idx, x, w, conv.bias, other, bias_other, y, geom, probing=True)      # noqa: E731
            best = _ck.choose((tag, geom), [(kind, geom, run)], two_kernels)
            plan = None if best is None else (best[1], geom, w)
        cache[key] = plan
    if plan is None:
        return two_kernels()
    idx, geom, w = plan
    y = torch.empty_like(other)
    _ck.conv(kind, idx, x, w, conv.bias, other, bias_other, y, geom)
    return y, None


def _site_input_grad_mask(g, conv, act, bits, other=None):
    """-> threshold_backward(conv's input gradient of ``g`` [+ other], act, 0): the ReLU in front of ``conv`` (``other``: the second
    branch of a residual junction).  ``bits``: act's pass bits where the forward left them.  The fused form, for any stride-1
    filter: composable_kernel's FORWARD kernel on the rewritten problem -- the input gradient of a stride-1 convolution is a forward
    convolution of ``g`` with the flipped, transposed filter -- with the threshold (and the junction add) as its epilogue."""
    def two_kernels():
        gx = _like(_conv_input_grad(g, act, conv), act)
        return _hip.relu_mask(gx, act, gx, gb=other, mask=bits)
    if not (_ck.enabled() and _dense_nhwc(g, act, other)):
        return two_kernels()
    tag = ("input_grad_mask", other is not None, bits is not None)
    kind = _ck.FWD_MASK if other is None else _ck.FWD_ADD_MASK
    plan, cache, key = _site(conv, tag, act.shape)
    if plan == "new":
        geom, plan = _ck.geometry(act.shape, conv), None
        fgeom = None if geom is None else _ck.backward_as_forward(geom)
        if fgeom is not None:
            wt = _ck.weight_flipped_cyxk(conv)
            out = torch.empty_like(act)
            d0, d1 = (act, None) if other is None else (other, act)
            best = _ck.choose((tag, geom), [(kind, fgeom, lambda idx: _ck.conv(kind, idx, g, wt, d0, d1, None, out, fgeom, probing=True))], two_kernels)
            plan = None if best is None else (best[1], fgeom, wt)
        cache[key] = plan
    if plan is None:
        return two_kernels()
    idx, fgeom, wt = plan
    gx = torch.empty_like(act)
    if other is None:
        _ck.conv(kind, idx, g, wt, act, None, None, gx, fgeom)
    else:
        _ck.conv(kind, idx, g, wt, other, act, None, gx, fgeom)
    return gx


def observed(module):
    """True if anybody hooked the module tree (then the plain module path must run: the hooks expect its calls)"""
    for m in module.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return True
    return False


def usable(net, x):
    """Can ``net`` (a backbones.resnet.ResNet) take the fused path for input ``x``?"""
    if os.environ.get("TA_FUSED_GLUE", "1") == "0" or net.training or x.dim() != 4 or x.dtype != torch.float32:
        return False
    if not getattr(net, "_bn_folded", False):
        return False
    if any(p.requires_grad for p in net.parameters()):
        return False                                    # only the input gradient is implemented
    return not observed(net) and topology_ok(net)


def topology_ok(net):
    """The fused forward hard-codes the topology of backbones.resnet (ReLU after every convolution, the stem's MaxPool2d, a
    mean-pool + fc head, a projection shortcut of one convolution).  A surrogate whose sub-modules were SWAPPED rather than
    hooked (a custom activation or pool for LinBP / BPA-style attacks, a ghost / dropout wrapper, an extra shortcut layer)
    must run as the modules say: exact types are checked on every call (a few dozen isinstance tests), anything else gets the
    module path."""
    from . import resnet
    nn = torch.nn
    if type(net) is not resnet.ResNet or type(net.relu) is not nn.ReLU or type(net.bn1) is not nn.Identity:
        return False
    mp = net.maxpool
    if type(mp) is not nn.MaxPool2d or _pair(mp.dilation) != [1, 1] or mp.ceil_mode or mp.return_indices:
        return False
    if type(net.avgpool) is not nn.AdaptiveAvgPool2d or net.avgpool.output_size not in (1, (1, 1)):
        return False
    if type(net.conv1) is not nn.Conv2d or type(net.fc) is not nn.Linear:
        return False
    kind = None
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        if type(layer) is not nn.Sequential:
            return False
        for blk in layer:
            if type(blk) not in (resnet.BasicBlock, resnet.Bottleneck) or (kind is not None and type(blk) is not kind):
                return False
            kind = type(blk)
            names = ("conv1", "conv2", "conv3") if kind is resnet.Bottleneck else ("conv1", "conv2")
            if set(dict(blk.named_children())) - set(names) - {"bn1", "bn2", "bn3", "relu", "downsample"}:
                return False
            if type(blk.relu) is not nn.ReLU:
                return False
            for cname in names:
                if type(getattr(blk, cname)) is not nn.Conv2d or type(getattr(blk, cname.replace("conv", "bn"))) is not nn.Identity:
                    return False
            ds = blk.downsample
            if ds is not None and not (isinstance(ds, nn.Sequential) and len(ds) == 2 and type(ds[0]) is nn.Conv2d
                                       and type(ds[1]) is nn.Identity):
                return False
    return True


def mark_folded(net):
    """called by backbones.fold_batchnorm: every Conv2d of the ResNet now carries its BatchNorm as a bias"""
    from . import resnet
    if not isinstance(net, resnet.ResNet):
        return
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    net._bn_folded = not bns and all(c.bias is not None for c in convs)


class _ResNetFn(torch.autograd.Function):
    """logits = net(x) with a hand-written input-gradient backward; the saved state is the post-ReLU activations (the
    threshold masks) -- the same tensors autograd would keep for the ReLUs, none of the pre-activation maps"""

    @staticmethod
    def forward(ctx, x, net, grad_scale=None):
        # grad_scale: the std vector by which the consumer of d(loss)/dx will divide it (the attack loop with the surrogate's
        # Normalize folded into its update, attack.py) -- the stem kernel then leaves the |dx / std[c]| sums with dx
        ctx.grad_scale = grad_scale
        bottleneck = hasattr(net.layer1[0], "conv3")
        if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
            x = x.contiguous()
        # pass bits (round 4): the backward reads an activation only for threshold_backward's sign test -- the forward glue
        # leaves that bit, the backward glue reads 1 bit instead of 4 bytes per element (TA_RELU_BITS=0: the activations)
        bits = os.environ.get("TA_RELU_BITS", "1") != "0"
        new_bits = (lambda t: _hip.pass_bits_like(t)) if bits else (lambda t: None)
        stem = _site_stem_bias_relu(x, net.conv1)
        # the stem pool: its own kernel where it is THE stem pool (3 x 3 / 2 / 1, NHWC) -- leaves a byte per pooled element for the
        # argmax and the pass bits of ``stem``; the backward then needs neither ATen's int64 indices nor ``stem`` itself
        if os.environ.get("TA_POOL_KERNEL", "1") != "0" and _hip.maxpool3s2_takes(stem, net.maxpool):
            pooled, idx, stem_bits = _hip.maxpool3s2_fwd(stem)
            stem = None
        else:
            pooled, idx = F.max_pool2d(stem, net.maxpool.kernel_size, net.maxpool.stride, net.maxpool.padding, return_indices=True)
            stem_bits = None
        saved, masks, cur = [], [], pooled
        for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
            for blk in layer:
                a, ma = _site_bias_relu(cur, blk.conv1, new_bits)
                if bottleneck:
                    b, mb = _site_bias_relu(a, blk.conv2, new_bits)
                    last_in, last = b, blk.conv3
                else:
                    b, mb, last_in, last = None, None, a, blk.conv2
                if blk.downsample is None:
                    y, my = _site_bias_add_relu(last_in, last, cur, None, new_bits)
                else:
                    y, my = _site_bias_add_relu(last_in, last, _conv(cur, blk.downsample[0]), blk.downsample[0].bias, new_bits)
                saved.append((a, b, y))
                masks.append((ma, mb, my))
                cur = y
        feat = cur.mean(dim=(2, 3))                                            # AdaptiveAvgPool2d(1) + flatten
        logits = F.linear(feat, net.fc.weight, net.fc.bias)
        # through save_for_backward: autograd frees the maps itself after a backward without retain_graph, a second backward
        # with retain_graph (vaifgsm.py:49 -- one gradient per auxiliary loss; adaea.py:44,51) finds them still there, and
        # an in-place write to a saved map by anybody is caught by the version check
        ctx.net, ctx.bottleneck = net, bottleneck
        flat = [x, stem, pooled, idx, stem_bits]
        for a, b, y in saved:
            flat += [a, b, y] if bottleneck else [a, y]
        for ma, mb, my in masks:          # pass bits per activation, None where there are none (TA_RELU_BITS=0, an odd size, a fused site)
            flat += [ma, mb, my] if bottleneck else [ma, my]
        ctx.save_for_backward(*flat)
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, g_logits):
        net, bottleneck = ctx.net, ctx.bottleneck
        flat = ctx.saved_tensors
        x, stem, pooled, idx, stem_bits = flat[:5]
        per = 3 if bottleneck else 2
        blocks = [blk for layer in (net.layer1, net.layer2, net.layer3, net.layer4) for blk in layer]
        end = 5 + per * len(blocks)
        saved = [(flat[j], flat[j + 1] if bottleneck else None, flat[j + per - 1]) for j in range(5, end, per)]
        masks = [(flat[j], flat[j + 1] if bottleneck else None, flat[j + per - 1]) for j in range(end, end + per * len(blocks), per)]
        last_y = saved[-1][2]
        n, c, h, w = last_y.shape
        g_feat = g_logits.mm(net.fc.weight)                                    # [N, C]
        # mean over H*W backward: expand(grad) / (H*W), as autograd's mean_backward
        g = _like(g_feat.view(n, c, 1, 1).expand(n, c, h, w) / (h * w), last_y)
        # gm: the gradient wrt a block's output AFTER that output's ReLU threshold.  For the last block it comes from the head; for
        # block i - 1 it is formed where block i's two branches meet -- threshold(conv1's input gradient + the shortcut's gradient):
        # one glue pass behind MIOpen's convolution, or that convolution's own epilogue (_site_input_grad_mask)
        gm = _hip.relu_mask(g, last_y, g, mask=masks[-1][2])
        g, pending = None, None
        for i in range(len(blocks) - 1, -1, -1):
            blk = blocks[i]
            a, b, y = saved[i]
            ma, mb, my = masks[i]
            x_in = saved[i - 1][2] if i > 0 else pooled
            if bottleneck:
                gb_ = _site_input_grad_mask(gm, blk.conv3, b, mb)
                ga_ = _site_input_grad_mask(gb_, blk.conv2, a, ma)
            else:
                ga_ = _site_input_grad_mask(gm, blk.conv2, a, ma)
            g_skip = gm if blk.downsample is None else _like(_conv_input_grad(gm, x_in, blk.downsample[0]), x_in)
            if i > 0:
                gm = _site_input_grad_mask(ga_, blk.conv1, x_in, masks[i - 1][2], other=g_skip)
            else:                                                              # the pooled map has no ReLU of its own: the junction
                g, pending = _like(_conv_input_grad(ga_, x_in, blk.conv1), x_in), g_skip      # is summed by the max-pool backward
        # the stem's ReLU sits before the max-pool: junction add + max-pool backward + threshold
        mp = net.maxpool
        k, st, pd = _pair(mp.kernel_size), _pair(mp.stride), _pair(mp.padding)
        cl = torch.channels_last
        if stem_bits is not None:                                              # the forward ran csrc/glue.hip's stem pool
            n_, c_, ph_, pw_ = pooled.shape
            g_stem = torch.empty((n_, c_, 2 * ph_, 2 * pw_), dtype=pooled.dtype, device=pooled.device, memory_format=cl)
            if not (g.is_contiguous(memory_format=cl) and pending.is_contiguous(memory_format=cl)):
                g, pending = g.contiguous(memory_format=cl), pending.contiguous(memory_format=cl)
            _hip.maxpool3s2_bwd_relu(g, idx, stem_bits, g_stem, gb=pending)
        elif (os.environ.get("TA_POOL_KERNEL", "1") != "0" and k[0] == k[1] and st[0] == st[1] and pd[0] == pd[1]
                and _pair(mp.dilation) == [1, 1] and not mp.ceil_mode and stem.shape[1] % 4 == 0
                and all(t.is_contiguous(memory_format=cl) and not t.is_contiguous() for t in (stem, idx, g, pending))):
            g_stem = _hip.maxpool_bwd_relu(g, idx, stem, torch.empty_like(stem), k[0], st[0], pd[0], gb=pending)
        else:
            g_pooled = g + pending
            g_stem = _like(torch.ops.aten.max_pool2d_with_indices_backward(g_pooled, stem, k, st, pd, [1, 1], False, idx), stem)
            _hip.relu_mask(g_stem, stem, g_stem)
        return _stem_input_grad(net, g_stem, x, ctx.grad_scale), None, None


def _stem_input_grad(net, g_stem, x, grad_scale=None):
    """d/d(image) through the stem convolution: the 7 x 7 / stride 2 / 3 -> 64 stem of the ResNets on the fp32-MFMA kernel
    (csrc/stem.hip; MIOpen's backward-data spends 7.5 % of a ResNet-50 iteration here), anything else through MIOpen"""
    conv = net.conv1
    if (os.environ.get("TA_STEM_KERNEL", "1") != "0" and tuple(conv.weight.shape) == (64, 3, 7, 7) and conv.stride == (2, 2)
            and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and x.shape[-1] % 2 == 0
            and x.shape[-2] % 2 == 0 and g_stem.is_contiguous(memory_format=torch.channels_last)):
        w2 = getattr(net, "_stem_w2", None)
        if w2 is None or w2.device != conv.weight.device or net._stem_w2_version != conv.weight._version:
            w2 = net._stem_w2 = _hip.stem7s2_prepare(conv.weight)
            net._stem_w2_version = conv.weight._version
        return _hip.stem7s2_input_grad(g_stem, w2, torch.empty(x.shape, dtype=x.dtype, device=x.device), std=grad_scale)
    return _conv_input_grad(g_stem, x, conv)


class _StemConvFn(torch.autograd.Function):
    """``conv1(x)`` of the plain MODULE path with the input gradient on csrc/stem.hip: the stem's backward-data is the last kernel of
    every surrogate backward and MIOpen's slowest (an implicit GEMM with a dimension of 3 input channels: 13 TFLOP/s), also -- in
    fact most of all -- in the reference-literal arrangement (NCHW, separate BatchNorm), where nothing else of this file applies.
    Forward = the module's own convolution; the kernel takes dy in either memory format, is deterministic, as accurate as MIOpen's
    (1.2e-6 vs 1.3e-6 of max|dx| from fp64), and leaves the sums of |dx / std| when the attack loop has folded the surrogate's
    Normalize into its update (attack.py) -- no sum-only pass before the update on this path either."""

    @staticmethod
    def forward(ctx, x, net, grad_scale):
        conv = net.conv1
        ctx.net, ctx.grad_scale, ctx.x_meta = net, grad_scale, (x.shape, x.dtype, x.device)
        return F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        shape, dtype, device = ctx.x_meta
        if not (g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last)):
            g = g.contiguous()
        net, conv = ctx.net, ctx.net.conv1
        w2 = getattr(net, "_stem_w2", None)
        if w2 is None or w2.device != conv.weight.device or net._stem_w2_version != conv.weight._version:
            w2 = net._stem_w2 = _hip.stem7s2_prepare(conv.weight)
            net._stem_w2_version = conv.weight._version
        return _hip.stem7s2_input_grad(g, w2, torch.empty(shape, dtype=dtype, device=device), std=ctx.grad_scale), None, None


def module_path_stem(net, x):
    """``net.conv1(x)`` -- through ``_StemConvFn`` when only the input gradient can be asked for (frozen weights), the stem has the
    7 x 7 / stride 2 / 3 -> 64 shape of csrc/stem.hip and nobody hooks ``conv1`` (module hooks expect the module's own call)."""
    conv = net.conv1
    if (os.environ.get("TA_STEM_KERNEL", "1") == "0" or not (torch.is_grad_enabled() and x.requires_grad) or not x.is_cuda
            or x.dtype != torch.float32 or x.dim() != 4 or type(conv) is not torch.nn.Conv2d
            or tuple(conv.weight.shape) != (64, 3, 7, 7) or conv.stride != (2, 2) or conv.padding != (3, 3) or conv.dilation != (1, 1)
            or conv.groups != 1 or conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)
            or x.shape[-1] % 2 or x.shape[-2] % 2 or observed(conv) or _global_module_hooks()):
        return conv(x)
    return _StemConvFn.apply(x, net, getattr(x, _hip._SCALE_ATTR, None))


def _global_module_hooks():
    from torch.nn.modules import module as m
    return any(getattr(m, name, None) for name in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks",
                                                   "_global_backward_pre_hooks", "_global_forward_hooks_always_called"))


def _like(t, ref):
    """``t`` in ``ref``'s dense memory format (a no-op when MIOpen already returned it that way)"""
    if ref.is_contiguous():
        return t.contiguous()
    return t.contiguous(memory_format=torch.channels_last)


def _pair(v):
    return [v, v] if isinstance(v, int) else list(v)


def forward(net, x):
    return _ResNetFn.apply(x, net, getattr(x, _hip._SCALE_ATTR, None))
