#!/usr/bin/env python
"""TIM conv alone (N=160 planes batch) for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip
_hip.load()
n = int(os.environ.get("TA_N", "160"))
g = [torch.randn(n, 3, 224, 224, device="cuda") for _ in range(2)]
o = torch.empty_like(g[0])
w = torch.rand(15, 15, device="cuda"); w = (w / w.sum()).contiguous()
for i in range(6):
    _hip.depthwise_conv2d_same(g[i % 2], o, w)
torch.cuda.synchronize()
print("done")
