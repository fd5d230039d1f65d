#!/usr/bin/env python
"""TIM smoothing alone: event timing of the direct convolution (the TA_TIM_VARIANT in the environment) and of the opt-in
separable form, at N = 32 and 160 images; also the launch set for rocprofv3 --pmc runs (TA_N picks one size)."""
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
    f = torch.rand(15, device="cuda")
    f = (f / f.sum()).contiguous()
    for name, call in (("direct (variant %s)" % os.environ.get("TA_TIM_VARIANT", "default"),
                        lambda i: _hip.depthwise_conv2d_same(g[i % 3], o, w)),
                       ("separable", lambda i: _hip.depthwise_conv2d_same_separable(g[i % 3], o, f, f))):
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
        print("n=%d %s: %.2f us  (%.2f TB/s at 8 B/element; direct-form FLOPs %.1f TFLOP/s)"
              % (n, name, us, elems * 8 / us / 1e6, elems * 450 / us / 1e6))
print("done")
