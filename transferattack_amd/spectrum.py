"""The spectrum transform of the reference's frequency-domain attacks (SSM: input_transformation/ssm.py:41-54, 101-209;
FGSRA: gradient/fgsra.py:49-140; both carry the DCT pair of the SSA repository as methods).

* ``spectrum_view(x, noise, mask)`` = ``idct_2d(dct_2d(x + noise) * mask)`` -- the expression both attacks evaluate per
  sample -- runs as two launches of the MFMA kernel ``ta_dct_pair`` (csrc/spectrum.hip): with the DCT-II matrix C and
  D = C^-1, ``y = D ((C (x + noise) C^T) * mask) D^T``.  Linear, so its backward is the same two launches with the
  transposed matrices.  Same transform as the reference's FFT factorisation, different rounding (~1e-6 relative).
* ``MakhoulDct`` keeps the reference's own factorisation (Makhoul, via torch.fft = rocFFT) for the ``dct`` / ``idct`` /
  ``dct_2d`` / ``idct_2d`` methods the attack classes expose, for plane sizes the kernel does not take (not a multiple of
  32, or above 256), and as the arithmetic the CPU tiers pin to the reference bit for bit."""
import math

import torch
from torch.autograd.function import once_differentiable

from . import _hip

_matrices = {}


def dct_matrices(n, device):
    """(C, D, C^T, D^T) fp32 on ``device``; C[k][m] = 2 cos(pi (2m + 1) k / (2n)) is the unnormalised DCT-II of the
    reference (``dct(x, norm=None)``), D its inverse (``idct``), both built in fp64"""
    key = (n, str(device))
    if key not in _matrices:
        k = torch.arange(n, dtype=torch.float64)[:, None]
        m = torch.arange(n, dtype=torch.float64)[None, :]
        forward = 2.0 * torch.cos(math.pi * (2.0 * m + 1.0) * k / (2.0 * n))
        inverse = torch.linalg.inv(forward)
        _matrices[key] = tuple(t.to(torch.float32).contiguous().to(device)
                               for t in (forward, inverse, forward.t(), inverse.t()))
    return _matrices[key]


def kernel_takes(x):
    return x.dim() == 4 and x.dtype == torch.float32 and x.shape[-1] == x.shape[-2] and x.shape[-1] % 32 == 0 \
        and 32 <= x.shape[-1] <= 256


class _SpectrumPair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise, mask):
        c, d, ct, dt = dct_matrices(x.shape[-1], x.device)
        x = x.contiguous()
        mid, y = torch.empty_like(x), torch.empty_like(x)
        _hip.dct_pair(x, noise, mask, mid, c, c)            # (C (x + noise) C^T) * mask
        _hip.dct_pair(mid, None, None, y, d, d)             # D . D^T
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (mask,) = ctx.saved_tensors
        c, d, ct, dt = dct_matrices(gy.shape[-1], gy.device)
        gy = gy.contiguous()
        mid, gx = torch.empty_like(gy), torch.empty_like(gy)
        _hip.dct_pair(gy, None, mask, mid, dt, dt)          # (D^T gy D) * mask
        _hip.dct_pair(mid, None, None, gx, ct, ct)          # C^T . C
        return gx, None, None


def spectrum_view(x, noise, mask):
    """idct_2d(dct_2d(x + noise) * mask); noise / mask: tensors shaped like x, no gradient"""
    if kernel_takes(x):
        return _SpectrumPair.apply(x, noise.contiguous(), mask.contiguous())
    pair = MakhoulDct()
    return pair.idct_2d(pair.dct_2d(x + noise) * mask)


class MakhoulDct:
    """DCT-II and its inverse by Makhoul's FFT factorisation, unnormalised; every step is differentiable.  The operation
    order follows the reference so that the CPU result is the reference's bit for bit."""

    def __init__(self):
        self._twiddles = {}

    def _cos_sin(self, n, like, sign):
        """cos / sin of -+ k*pi/(2N), k < N, kept per (N, device, dtype)"""
        key = (n, like.device, like.dtype, sign)
        if key not in self._twiddles:
            ramp = torch.arange(n, dtype=like.dtype, device=like.device)[None, :]
            angle = (-ramp if sign < 0 else ramp) * math.pi / (2 * n)
            self._twiddles[key] = (torch.cos(angle), torch.sin(angle))
        return self._twiddles[key]

    def dct(self, x, norm=None):
        if norm is not None:
            raise Exception("Unsupported DCT normalisation {}".format(norm))
        shape, n = x.shape, x.shape[-1]
        rows = x.contiguous().view(-1, n)
        folded = torch.cat([rows[:, ::2], rows[:, 1::2].flip([1])], dim=1)       # even samples, then odd reversed
        spectrum = torch.fft.fft(folded)
        cos_k, sin_k = self._cos_sin(n, rows, -1)
        out = spectrum.real * cos_k - spectrum.imag * sin_k
        return 2 * out.view(*shape)

    def idct(self, X, norm=None):
        if norm is not None:
            raise Exception("Unsupported DCT normalisation {}".format(norm))
        shape, n = X.shape, X.shape[-1]
        re = X.contiguous().view(-1, n) / 2
        im = torch.cat([re[:, :1] * 0, -re.flip([1])[:, :-1]], dim=1)
        cos_k, sin_k = self._cos_sin(n, re, +1)
        rotated = torch.complex(re * cos_k - im * sin_k, re * sin_k + im * cos_k)
        folded = torch.fft.ifft(rotated)
        rows = folded.new_zeros(folded.shape)
        rows[:, ::2] += folded[:, :n - (n // 2)]
        rows[:, 1::2] += folded.flip([1])[:, :n // 2]
        return rows.view(*shape).real

    def dct_2d(self, x, norm=None):
        return self.dct(self.dct(x, norm=norm).transpose(-1, -2), norm=norm).transpose(-1, -2)

    def idct_2d(self, x, norm=None):
        return self.idct(self.idct(x, norm=norm).transpose(-1, -2), norm=norm).transpose(-1, -2)
