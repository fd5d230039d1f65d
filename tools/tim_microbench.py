#!/usr/bin/env python
"""TIM smoothing and the DIM kernels alone: event timing at N = 32 and 160 images; also the launch set for rocprofv3 --pmc
runs (TA_N picks one size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

_hip.load()
sizes = [int(os.environ["TA_N"])] if "TA_N" in os.environ else [32, 160]
for n in sizes:
    g = [torch.randn(n, 3, 224, 224, device="cuda") for _ in range(3)]
    o = torch.empty_like(g[0])
    w = torch.rand(15, 15, device="cuda")
    w = (w / w.sum()).contiguous()
    fwd = lambda i: _hip.dim_fwd(g[i % 3], o, 246, 237, 3, 5)      # noqa: E731
    bwd = lambda i: _hip.dim_bwd(g[i % 3], o, 246, 237, 3, 5)      # noqa: E731
    cases = (("tim 15x15", lambda i: _hip.depthwise_conv2d_same(g[i % 3], o, w)), ("dim_fwd", fwd), ("dim_bwd", bwd))
    for name, call in cases:
        for i in range(6):
            call(i)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for i in range(20):
            call(i)
        end.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(end) * 1e3 / 20
        elems = n * 3 * 224 * 224
        print("n=%d %s: %.2f us  (%.2f TB/s at 8 B/element%s)"
              % (n, name, us, elems * 8 / us / 1e6, "; %.1f TFLOP/s" % (elems * 450 / us / 1e6) if "tim" in name else ""))
print("done")
