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
