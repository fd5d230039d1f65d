"""TEST INFRASTRUCTURE: an oracle-backed stand-in for ``transferattack_amd._hip`` so the *host logic* of the
attack classes (hook plumbing, RNG draw order, autograd wiring, sharding) can be exercised on the CPU box.

The product never sees this: ``install(monkeypatch)`` swaps the functions of the binding module for the
duration of one test.  Every function has the signature of its ``_hip`` counterpart and is implemented with
the oracle (oracle/fgsm_oracle.py, oracle/ta_oracle.c) on CPU tensors.
"""
import numpy as np
import torch
import torch.nn.functional as F

import c_oracle as C
import fgsm_oracle as O
from transferattack_amd import _hip

calls = []


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def momentum(grad, momentum_in, momentum_out, decay, variance=None):
    calls.append("momentum")
    g = grad if variance is None else grad + variance
    momentum_out.copy_(O.momentum_step(g, 0 if momentum_in is None else momentum_in, decay))


def update_delta_linf(delta_in, data, momentum_, alpha, epsilon, delta_out, x_adv=None):
    calls.append("update_delta_linf")
    delta_out.copy_(O.delta_step(delta_in, data, momentum_, alpha, epsilon))
    if x_adv is not None:
        x_adv.copy_(data + delta_out)


def update_delta_l2(delta_in, data, grad, alpha, epsilon, delta_out):
    calls.append("update_delta_l2")
    delta_out.copy_(O.delta_step(delta_in, data, grad, alpha, epsilon, norm="l2"))


def u8_source_probe(data):
    calls.append("u8_source_probe")
    u8 = torch.round(data * 255).clamp(0, 255).to(torch.uint8)
    exact = bool(torch.equal(u8.float() / 255, data))
    return u8, torch.tensor([0 if exact else 1], dtype=torch.int32)


def mi_update(grad, momentum_in, momentum_out, delta, data, decay, alpha, epsilon, variance=None, x_adv=None, data_u8=None,
              std=None):
    calls.append("mi_update" if std is None else "mi_update_std")
    if data_u8 is not None and int(data_u8[1]) == 0:            # the byte source stands for exactly these floats
        assert torch.equal(data_u8[0].float() / 255, data)
    if std is not None:                                         # Normalize's backward folded into the update
        assert variance is None and x_adv is None
        grad = grad / std.view(1, -1, 1, 1)
    g = grad if variance is None else grad + variance
    m = O.momentum_step(g, 0 if momentum_in is None else momentum_in, decay)
    d = O.delta_step(delta, data, m, alpha, epsilon)
    if momentum_out is not None:
        momentum_out.copy_(m)
    delta.copy_(d)
    if x_adv is not None:
        x_adv.copy_(data + d)


def resize_normalize_fwd(x, y, mean, std):
    calls.append("resize_normalize_fwd")
    v = torch.nn.functional.interpolate(x, size=tuple(y.shape[-2:]), mode="bilinear", align_corners=False)
    y.copy_((v - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1))


def resize_normalize_bwd(gy, gx, std):
    calls.append("resize_normalize_bwd")
    with torch.enable_grad():
        xin = torch.zeros_like(gx).requires_grad_(True)
        v = torch.nn.functional.interpolate(xin, size=tuple(gy.shape[-2:]), mode="bilinear", align_corners=False)
        gx.copy_(torch.autograd.grad(v, xin, gy / std.view(1, -1, 1, 1))[0])


def init_delta_uniform(delta, data, epsilon, seed=0, offset=0, noise=None):
    calls.append("init_delta_uniform")
    if noise is None:
        noise = _t(C.philox_uniform(delta.numel(), seed, offset, epsilon)).view_as(delta)
    delta.copy_(O.box_clamp(noise, 0 - data, 1 - data))


def depthwise_conv2d_same(inp, out, weight2d):
    calls.append("depthwise_conv2d_same")
    c = inp.shape[1]
    out.copy_(F.conv2d(inp, weight2d[None, None].repeat(c, 1, 1, 1), stride=1, padding="same", groups=c))


def dim_fwd(x, y, resize, rnd, top, left):
    calls.append("dim_fwd")
    y.copy_(_t(C.dim_fwd(x.numpy(), (True, rnd, top, left), resize)))


def dim_bwd(gy, gx, resize, rnd, top, left):
    calls.append("dim_bwd")
    gx.copy_(_t(C.dim_bwd(gy.numpy(), (True, rnd, top, left), resize)))


def scale_copies_fwd(x, y, num_scale):
    calls.append("scale_copies_fwd")
    y.copy_(O.sim_copies(x, num_scale))


def scale_copies_bwd(gy, gx, num_scale):
    calls.append("scale_copies_bwd")
    with torch.enable_grad():
        xin = torch.zeros_like(gx, requires_grad=True)
        res = torch.autograd.grad(O.sim_copies(xin, num_scale), xin, gy)[0]
    gx.copy_(res)


def sum_copies_bwd(gy, gx, copies):
    calls.append("sum_copies_bwd")
    parts = gy.view((copies,) + tuple(gx.shape))
    acc = parts[copies - 1].clone()
    for i in range(copies - 2, -1, -1):
        acc = acc + parts[i]
    gx.copy_(acc)


def admix_fwd(x, perm, y, num_admix, num_scale, strength):
    calls.append("admix_fwd")
    y.copy_(O.admix_copies(x, list(perm.view(num_admix, -1)), strength, num_scale))


def admix_bwd(gy, gx, num_admix, num_scale):
    calls.append("admix_bwd")
    n = gx.shape[0]
    with torch.enable_grad():
        xin = torch.zeros_like(gx, requires_grad=True)
        perms = [torch.arange(n) for _ in range(num_admix)]
        res = torch.autograd.grad(O.admix_copies(xin, perms, 0.2, num_scale), xin, gy)[0]
    gx.copy_(res)


def vmi_neighbor(data, delta, out, radius, seed=0, offset=0, noise=None):
    calls.append("vmi_neighbor")
    if noise is None:
        noise = _t(C.philox_uniform(data.numel(), seed, offset, radius)).view_as(data)
    out.copy_(data + delta + noise)


def grad_accumulate(acc, grad, first):
    calls.append("grad_accumulate")
    if first:
        acc.copy_(grad)
    else:
        acc.add_(grad)


def variance_finalize(acc, cur_grad, out, count):
    calls.append("variance_finalize")
    out.copy_(acc / count - cur_grad)


def axpy(x, m, coeff, out):
    calls.append("axpy")
    out.copy_(x + coeff * m)


def normalize_adv_fwd(data, delta, y, mean, std, data_u8=None):
    calls.append("normalize_adv_fwd")
    if data_u8 is not None and int(data_u8[1]) == 0:
        assert torch.equal(data_u8[0].float() / 255, data)
    y.copy_(((data + delta) - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1))


def normalize_fwd(x, y, mean, std):
    calls.append("normalize_fwd")
    y.copy_((x - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1))


def normalize_bwd(gy, gx, std, variance=None):
    calls.append("normalize_bwd")
    gx.copy_(gy / std.view(1, -1, 1, 1))


def vmi_neighbor_normalized(data, delta, out, mean, std, radius, seed=0, offset=0, noise=None):
    calls.append("vmi_neighbor_normalized")
    if noise is None:
        noise = _t(C.philox_uniform(data.numel(), seed, offset, radius)).view_as(data)
    out.copy_((((data + delta) + noise) - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1))


def normalize_bwd_accumulate(gy, acc, std, first):
    calls.append("normalize_bwd_accumulate")
    g = gy / std.view(1, -1, 1, 1)
    if first:
        acc.copy_(g)
    else:
        acc.add_(g)


def quantize_u8_nhwc(data, delta, out):
    calls.append("quantize_u8_nhwc")
    out.copy_(_t(O.quantize_u8(data + delta)))


def sum_members(grads, gx):
    calls.append("sum_members")
    acc = grads[-1].clone()
    for g in reversed(grads[:-1]):
        acc = acc + g
    gx.copy_(acc)


def bsr_fwd(x, plan, y, copies, num_block):
    calls.append("bsr_fwd")
    y.copy_(O.bsr_apply_table(x, plan.cpu().numpy(), num_block))


def bsr_bwd(gy, plan, gx, copies, num_block):
    calls.append("bsr_bwd")
    with torch.enable_grad():
        xin = torch.zeros_like(gx, requires_grad=True)
        gx.copy_(torch.autograd.grad(O.bsr_apply_table(xin, plan.cpu().numpy(), num_block), xin, gy)[0])


def _sia_plans(plan, num_block, n, noise):
    """decode the int32 plan table of transforms.sia_draw back into the oracle's per-copy dictionaries"""
    import struct
    plans = []
    for k, row in enumerate(plan.cpu().numpy()):
        rows, cols = row[:num_block + 1].tolist(), row[num_block + 1:2 * (num_block + 1)].tolist()
        blocks, cell = [], 2 * (num_block + 1)
        for i in range(num_block):
            for j in range(num_block):
                op, step, bits = (int(v) for v in row[cell:cell + 3])
                cell += 3
                scale = torch.tensor(struct.unpack("<f", struct.pack("<i", bits))[0], dtype=torch.float32)
                nz = None
                if op == 6:
                    nz = noise[k * n:(k + 1) * n, :, rows[i]:rows[i + 1], cols[j]:cols[j + 1]]
                blocks.append((op, step, scale, nz))
        plans.append(dict(rows=rows, cols=cols, blocks=blocks))
    return plans


def sia_fwd(x, plan, y, copies, num_block, noise_radius, seed=0, offset=0, noise=None):
    calls.append("sia_fwd")
    assert noise is not None, "the fake has no Philox stream for SIA: inject the noise"
    y.copy_(O.sia_apply(x, _sia_plans(plan, num_block, x.shape[0], noise)))


def sia_bwd(gy, plan, x, gx, copies, num_block, noise_radius, seed=0, offset=0, noise=None):
    calls.append("sia_bwd")
    with torch.enable_grad():
        xin = x.detach().clone().requires_grad_(True)
        y = O.sia_apply(xin, _sia_plans(plan, num_block, x.shape[0], noise))
        gx.copy_(torch.autograd.grad(y, xin, gy)[0])


_NAMES = ["sia_fwd", "sia_bwd", "bsr_fwd", "bsr_bwd", "sum_members", "momentum", "update_delta_linf", "update_delta_l2", "mi_update", "u8_source_probe", "init_delta_uniform",
          "depthwise_conv2d_same", "dim_fwd", "dim_bwd", "scale_copies_fwd", "scale_copies_bwd", "sum_copies_bwd", "admix_fwd",
          "admix_bwd", "vmi_neighbor", "grad_accumulate", "variance_finalize", "axpy", "quantize_u8_nhwc", "normalize_fwd",
          "normalize_bwd", "normalize_adv_fwd", "vmi_neighbor_normalized", "normalize_bwd_accumulate", "resize_normalize_fwd", "resize_normalize_bwd"]


def _fft_spectrum_view(x, noise, mask):
    """the reference's own arithmetic (Makhoul FFT factorisation on torch's CPU path): what the host-logic tier pins"""
    from transferattack_amd.spectrum import MakhoulDct
    calls.append("spectrum_view")
    pair = MakhoulDct()
    return pair.idct_2d(pair.dct_2d(x + noise) * mask)


def install(monkeypatch):
    del calls[:]
    from transferattack_amd import spectrum
    monkeypatch.setattr(spectrum, "spectrum_view", _fft_spectrum_view)
    for name in _NAMES:
        assert hasattr(_hip, name), name
        monkeypatch.setattr(_hip, name, globals()[name])
