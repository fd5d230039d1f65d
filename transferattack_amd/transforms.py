"""Differentiable input transforms of the FGSM family as ``torch.autograd.Function`` s whose forward and
backward are single HIP kernels (the reference builds them from 3-20 ATen ops inside the autograd graph):

    DimResizePad   DIM.transform   input_transformation/dim.py:42-68
    ScaleCopies    SIM.transform   input_transformation/sim.py:36-40
    AdmixCopies    Admix.transform input_transformation/admix.py:40-45
    LookAhead      NIFGSM.transform gradient/nifgsm.py:35-39
    Neighbor       VMI sampling    gradient/vmifgsm.py:50
    SiaBlocks      SIA.transform   input_transformation/sia.py:86-100
    BsrBlocks      BSR.transform   input_transformation/bsr.py:57-67

The random draws stay on the host, on torch's CPU default generator, in the reference's order, so a seeded
run makes the same choices as the reference on any device.
"""
import struct

import numpy as np
import torch
from torch.autograd.function import once_differentiable

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
    @once_differentiable
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
    @once_differentiable
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
    @once_differentiable
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
    @once_differentiable
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
    @once_differentiable
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
    @once_differentiable
    def backward(ctx, gy):
        x, plan, noise = ctx.saved_tensors
        copies, num_block, seed, offset = ctx.cfg
        gx = torch.empty_like(x)
        _hip.sia_bwd(gy.contiguous(), plan, x, gx, copies, num_block, SIA_NOISE, seed, offset, noise)
        return gx, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ BSR
BSR_DEGREES = 24.0           # bsr.py:54: RandomRotation(degrees=(-24, 24))


def _bsr_lengths(length, num_block):
    """BSR.get_length (bsr.py:41-45)"""
    rand = np.random.uniform(2, size=num_block)
    rand_norm = np.round(rand / rand.sum() * length).astype(np.int32)
    rand_norm[rand_norm.argmax()] += length - rand_norm.sum()
    return [int(v) for v in rand_norm]


def bsr_draw(shape, num_block, num_copies):
    """The random choices of ``num_copies`` calls of BSR.shuffle on a batch of ``shape`` (bsr.py:47-61), drawn in the
    reference's order from the reference's three host generators: python ``random`` -- the order of the two axes and the
    strip / block permutations; numpy -- the strip lengths; torch (CPU) -- one rotation angle per strip
    (``torch.empty(1).uniform_(-24, 24)``, what torchvision's RandomRotation draws).  Returns the int32 plan table
    ``ta_bsr_fwd`` reads ([copies, 1 + 7*nb + 3*nb*nb], strips and blocks in OUTPUT order); the rotation enters as the
    four entries of theta^T / (w/2, h/2) in fp32 (torchvision's _gen_affine_grid), computed like the reference computes
    them: the matrix in double precision, rounded to fp32, divided in fp32."""
    import math
    import random
    _, _, height, width = shape
    size = (height, width)
    nb = num_block
    stride = 1 + 7 * nb + 3 * nb * nb
    plan = np.zeros((num_copies, stride), dtype=np.int32)
    as_bits = lambda v: int(np.float32(v).view(np.int32))         # noqa: E731
    for k in range(num_copies):
        dims = [2, 3]
        random.shuffle(dims)
        first, second = dims[0] - 2, dims[1] - 2                   # 0 = rows, 1 = columns
        lengths0 = _bsr_lengths(size[first], nb)
        order0 = list(range(nb))
        random.shuffle(order0)
        starts0 = np.concatenate([[0], np.cumsum(lengths0)[:-1]])
        plan[k, 0] = first
        out0 = 0
        for pos, src in enumerate(order0):
            angle = float(torch.empty(1).uniform_(-BSR_DEGREES, BSR_DEGREES).item())
            lengths1 = _bsr_lengths(size[second], nb)
            order1 = list(range(nb))
            random.shuffle(order1)
            h = lengths0[src] if first == 0 else height           # the strip tensor is h x w
            w = width if first == 0 else lengths0[src]
            rot = math.radians(-angle)                             # functional.rotate inverts the angle
            m = np.array([math.cos(rot), math.sin(rot), -math.sin(rot), math.cos(rot)], dtype=np.float32)
            half_w, half_h = np.float32(0.5 * w), np.float32(0.5 * h)
            rt = (m[0] / half_w, m[1] / half_w, m[2] / half_h, m[3] / half_h)          # rt00, rt10, rt01, rt11
            base = 1 + 7 * pos
            plan[k, base:base + 3] = (starts0[src], lengths0[src], out0)
            plan[k, base + 3:base + 7] = [as_bits(v) for v in rt]
            out0 += lengths0[src]
            starts1 = np.concatenate([[0], np.cumsum(lengths1)[:-1]])
            out1 = 0
            for j, blk in enumerate(order1):
                cell = 1 + 7 * nb + 3 * (nb * pos + j)
                plan[k, cell:cell + 3] = (starts1[blk], lengths1[blk], out1)
                out1 += lengths1[blk]
    return plan


class BsrBlocks(torch.autograd.Function):
    """cat of ``copies`` shuffled-and-rotated clones of x: one gather kernel each way (``ta_bsr_fwd/bwd``)."""

    @staticmethod
    def forward(ctx, x, plan, copies, num_block):
        x = x.contiguous()
        y = torch.empty((copies * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _hip.bsr_fwd(x, plan, y, copies, num_block)
        ctx.save_for_backward(plan)
        ctx.cfg = (copies, num_block, tuple(x.shape))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (plan,) = ctx.saved_tensors
        copies, num_block, shape = ctx.cfg
        gx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        _hip.bsr_bwd(gy.contiguous(), plan, gx, copies, num_block)
        return gx, None, None, None
