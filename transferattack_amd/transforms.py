"""Differentiable input transforms of the FGSM family as ``torch.autograd.Function`` s whose forward and
backward are single HIP kernels (the reference builds them from 3-20 ATen ops inside the autograd graph):

    DimResizePad   DIM.transform   input_transformation/dim.py:42-68
    ScaleCopies    SIM.transform   input_transformation/sim.py:36-40
    AdmixCopies    Admix.transform input_transformation/admix.py:40-45
    LookAhead      NIFGSM.transform gradient/nifgsm.py:35-39
    Neighbor       VMI sampling    gradient/vmifgsm.py:50

The random draws stay on the host, on torch's CPU default generator, in the reference's order, so a seeded
run makes the same choices as the reference on any device.
"""
import struct

import numpy as np
import torch

from . import _hip


def dim_draw(img_size, resize_rate, diversity_prob):
    """One DIM geometry per call for the whole batch (dim.py:47-63): rand -> randint(rnd) -> randint(top) ->
    randint(left); returns None when the transform is skipped (probability 1 - diversity_prob)."""
    if torch.rand(1) > diversity_prob:
        return None
    img_resize = int(img_size * resize_rate)
    rnd = int(torch.randint(low=min(img_size, img_resize), high=max(img_size, img_resize), size=(1,),
                            dtype=torch.int32))
    rem = img_resize - rnd
    top = int(torch.randint(low=0, high=rem, size=(1,), dtype=torch.int32))
    left = int(torch.randint(low=0, high=rem, size=(1,), dtype=torch.int32))
    return img_resize, rnd, top, left


class DimResizePad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, resize, rnd, top, left):
        x = x.contiguous()
        y = torch.empty_like(x)
        _hip.dim_fwd(x, y, resize, rnd, top, left)
        ctx.geom = (resize, rnd, top, left)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        _hip.dim_bwd(gy, gx, *ctx.geom)
        return gx, None, None, None, None


class ScaleCopies(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, num_scale):
        x = x.contiguous()
        y = torch.empty((num_scale * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _hip.scale_copies_fwd(x, y, num_scale)
        ctx.num_scale = num_scale
        ctx.in_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty(ctx.in_shape, dtype=gy.dtype, device=gy.device)
        _hip.scale_copies_bwd(gy, gx, ctx.num_scale)
        return gx, None


class AdmixCopies(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, perm, num_admix, num_scale, strength):
        x = x.contiguous()
        y = torch.empty((num_scale * num_admix * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _hip.admix_fwd(x, perm, y, num_admix, num_scale, strength)
        ctx.cfg = (num_admix, num_scale)
        ctx.in_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty(ctx.in_shape, dtype=gy.dtype, device=gy.device)
        _hip.admix_bwd(gy, gx, *ctx.cfg)
        return gx, None, None, None, None


class LookAhead(torch.autograd.Function):
    """x + coeff * momentum; the momentum carries no gradient (it is detached state)."""

    @staticmethod
    def forward(ctx, x, momentum, coeff):
        x = x.contiguous()
        out = torch.empty_like(x)
        _hip.axpy(x, momentum.contiguous(), coeff, out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class Neighbor(torch.autograd.Function):
    """data + delta + U(-radius, radius): in-kernel Philox (seed, offset) or injected ``noise``; the
    gradient flows to ``delta`` unchanged."""

    @staticmethod
    def forward(ctx, delta, data, radius, seed, offset, noise):
        out = torch.empty_like(data)
        _hip.vmi_neighbor(data, delta.contiguous(), out, radius, seed, offset, noise)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ SIA
SIA_OPS = 7                 # roll rows, roll columns, flip rows, flip columns, rotate 180, scale, noise + clip
SIA_NOISE = 16 / 255        # sia.py:67: a constant of the method, not the attack's epsilon


def sia_draw(shape, num_block, num_copies, noise_source=None):
    """The random choices of ``num_copies`` block transforms of a batch of ``shape`` (sia.py:86-100), drawn in the
    reference's order from the reference's host generators: numpy -- column cuts, row cuts, then per rectangle (rows
    outer) the operation and, for the two rolls, the step; torch (CPU) -- the scale factor.  Returns the int32 plan
    table ``ta_sia_fwd`` reads ([copies, 2*(nb+1) + 3*nb*nb]) and, if ``noise_source`` is given (tests: the
    reference's CPU draws, in the reference's order), the noise as a tensor shaped like the output stack; otherwise
    None and the kernel draws from its Philox stream."""
    n, c, height, width = shape
    stride = 2 * (num_block + 1) + 3 * num_block * num_block
    plan = np.zeros((num_copies, stride), dtype=np.int32)
    noise = torch.zeros((num_copies * n, c, height, width)) if noise_source is not None else None
    for k in range(num_copies):
        cols = [0] + np.random.choice(list(range(1, width)), num_block - 1, replace=False).tolist() + [width]
        rows = [0] + np.random.choice(list(range(1, height)), num_block - 1, replace=False).tolist() + [height]
        cols.sort()
        rows.sort()
        plan[k, :num_block + 1] = rows
        plan[k, num_block + 1:2 * (num_block + 1)] = cols
        cell = 2 * (num_block + 1)
        for i in range(num_block):
            for j in range(num_block):
                bh, bw = rows[i + 1] - rows[i], cols[j + 1] - cols[j]
                op = int(np.random.randint(0, high=SIA_OPS, dtype=np.int32))
                step, scale_bits = 0, 0
                if op == 0:
                    step = int(np.random.randint(low=0, high=bh, dtype=np.int32))
                elif op == 1:
                    step = int(np.random.randint(low=0, high=bw, dtype=np.int32))
                elif op == 5:
                    scale_bits = struct.unpack("<i", struct.pack("<f", float(torch.rand(1)[0])))[0]
                elif op == 6 and noise is not None:
                    noise[k * n:(k + 1) * n, :, rows[i]:rows[i + 1], cols[j]:cols[j + 1]] = noise_source(
                        (n, c, bh, bw), -SIA_NOISE, SIA_NOISE)
                plan[k, cell:cell + 3] = (op, step, scale_bits)
                cell += 3
    return plan, noise


class SiaBlocks(torch.autograd.Function):
    """cat of ``copies`` block-transformed clones of x: one gather kernel each way (``ta_sia_fwd/bwd``)."""

    @staticmethod
    def forward(ctx, x, plan, copies, num_block, seed, offset, noise):
        x = x.contiguous()
        y = torch.empty((copies * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _hip.sia_fwd(x, plan, y, copies, num_block, SIA_NOISE, seed, offset, noise)
        ctx.save_for_backward(x, plan, noise)
        ctx.cfg = (copies, num_block, seed, offset)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, plan, noise = ctx.saved_tensors
        copies, num_block, seed, offset = ctx.cfg
        gx = torch.empty_like(x)
        _hip.sia_bwd(gy.contiguous(), plan, x, gx, copies, num_block, SIA_NOISE, seed, offset, noise)
        return gx, None, None, None, None, None, None
