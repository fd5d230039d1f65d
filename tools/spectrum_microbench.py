#!/usr/bin/env python
"""The spectrum view y = idct_2d(dct_2d(x + noise) * mask) on MI355X: the two-launch fp32-MFMA kernel (ta_dct_pair) against
the reference's FFT factorisation on torch.fft (rocFFT) and against the same products through torch.matmul (rocBLAS)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip, spectrum  # noqa: E402

_hip.load()
for n_img in (16, 32, 125):
    shape = (n_img, 3, 224, 224)
    x, noise, mask = (torch.rand(shape, device="cuda") for _ in range(3))
    fft = spectrum.MakhoulDct()
    c, d, ct, dt = spectrum.dct_matrices(224, x.device)
    forms = (("MFMA kernel (2 launches)", lambda: spectrum.spectrum_view(x, noise, mask)),
             ("rocBLAS (4 matmuls + 2 elementwise)", lambda: d @ ((c @ (x + noise) @ ct) * mask) @ dt),
             ("rocFFT factorisation (reference's form)", lambda: fft.idct_2d(fft.dct_2d(x + noise) * mask)))
    for name, fn in forms:
        for _ in range(3):
            fn()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for _ in range(10):
            fn()
        end.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(end) * 1e3 / 10
        elems = x.numel()
        print("n=%d %-42s %9.1f us   %6.1f TFLOP/s at 1792 FLOP/element, %5.0f GB/s at 16 B/element"
              % (n_img, name, us, elems * 1792 / us / 1e6, elems * 16 / us / 1e3))
print("done")
