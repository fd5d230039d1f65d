"""DCT-II and its inverse by Makhoul's FFT factorisation, unnormalised -- the transform pair the reference's
frequency-domain attacks carry as methods (gradient/fgsra.py:49-123, input_transformation/ssm.py:101-209; both copied
there from the SSA repository).  torch.fft is rocFFT on the device; every step is differentiable.  The operation
order follows the reference so that the CPU result is the reference's bit for bit."""
import math
import os

import torch


class MakhoulDct:
    """``TA_DCT_GEMM=1`` (opt-in) evaluates the 2-D pair as dense products with the N x N DCT matrix instead --
    Y = C X C^T, X = D Y D^T with D = C^-1 (built in fp64, cast to fp32) -- i.e. four plain GEMMs (rocBLAS, MFMA) in
    place of ~60 small kernels and two non-power-of-two FFTs per transform.  Same transform, different rounding
    (~1e-6 relative), so it is not the default; still differentiable."""

    def __init__(self):
        self._twiddles = {}
        self._matrices = {}
        self.use_gemm = os.environ.get("TA_DCT_GEMM", "0") == "1"

    def _dct_matrices(self, n, like):
        """(C, C^-1) with C[k][m] = 2 cos(pi (2m + 1) k / (2n)): the unnormalised DCT-II of this module as a matrix"""
        key = (n, like.device, like.dtype)
        if key not in self._matrices:
            k = torch.arange(n, dtype=torch.float64)[:, None]
            m = torch.arange(n, dtype=torch.float64)[None, :]
            forward = 2.0 * torch.cos(math.pi * (2.0 * m + 1.0) * k / (2.0 * n))
            inverse = torch.linalg.inv(forward)
            self._matrices[key] = (forward.to(like.dtype).to(like.device), inverse.to(like.dtype).to(like.device))
        return self._matrices[key]

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
        if self.use_gemm and norm is None:
            rows, cols = self._dct_matrices(x.shape[-2], x)[0], self._dct_matrices(x.shape[-1], x)[0]
            return rows @ x @ cols.transpose(0, 1)
        return self.dct(self.dct(x, norm=norm).transpose(-1, -2), norm=norm).transpose(-1, -2)

    def idct_2d(self, x, norm=None):
        if self.use_gemm and norm is None:
            rows, cols = self._dct_matrices(x.shape[-2], x)[1], self._dct_matrices(x.shape[-1], x)[1]
            return rows @ x @ cols.transpose(0, 1)
        return self.idct(self.idct(x, norm=norm).transpose(-1, -2), norm=norm).transpose(-1, -2)
