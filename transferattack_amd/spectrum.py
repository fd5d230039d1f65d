"""DCT-II and its inverse by Makhoul's FFT factorisation, unnormalised -- the transform pair the reference's
frequency-domain attacks carry as methods (gradient/fgsra.py:49-123, input_transformation/ssm.py:101-209; both copied
there from the SSA repository).  torch.fft is rocFFT on the device; every step is differentiable.  The operation
order follows the reference so that the CPU result is the reference's bit for bit."""
import math

import torch


class MakhoulDct:
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
