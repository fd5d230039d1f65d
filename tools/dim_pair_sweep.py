#!/usr/bin/env python
"""DIM lane kernels: tiles per workgroup whose loads are all issued up front (TA_DIM_PAIR = 0 (one-tile kernels of rounds 2-4),
1, 2, 4), event timing at N = 32 and 160."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

_hip.load()
for n in (32, 160):
    g = [torch.randn(n, 3, 224, 224, device="cuda") for _ in range(3)]
    o = torch.empty_like(g[0])
    for what, fn in (("dim_fwd", lambda i: _hip.dim_fwd(g[i % 3], o, 246, 237, 3, 5)),
                     ("dim_bwd", lambda i: _hip.dim_bwd(g[i % 3], o, 246, 237, 3, 5))):
        row = []
        for pair in os.environ.get("TA_SWEEP", "0,1,2,4").split(","):
            os.environ["TA_DIM_PAIR"] = pair
            for i in range(6):
                fn(i)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for i in range(30):
                fn(i)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / 30
            row.append("PAIR=%s %.2f us (%.2f TB/s)" % (pair, us, n * 150528 * 8 / us / 1e6))
        print("n=%d %s: %s" % (n, what, "   ".join(row)), flush=True)
