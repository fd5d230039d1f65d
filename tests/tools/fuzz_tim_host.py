"""Fuzz the depthwise-convolution kernels (host stand-in, tests/hipcpu) against the C oracle.
    python tests/tools/fuzz_tim_host.py <seed> <cases>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle', ROOT + '/tests'):
    sys.path.insert(0, p)
import c_oracle as C          # noqa: E402
import host_kernels           # noqa: E402


class P:
    def setattr(self, o, n, v):
        setattr(o, n, v)

    def setenv(self, n, v):
        os.environ[n] = v


host_kernels.install(P(), tag='timfuzz', env={})
from transferattack_amd import _hip   # noqa: E402
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    k = int(rng.choice([3, 5, 7, 15, 4, 9, 11]))
    fast = rng.rand() < 0.6
    w = int(rng.randint(1, 57)) * 4 if fast else int(rng.randint(3, 300))
    h = int(rng.randint(3, 260))
    planes = int(rng.randint(1, 4))
    g = torch.randn(1, planes, h, w)
    wt = torch.rand(k, k)
    wt = (wt / wt.sum()).contiguous()
    out = torch.empty_like(g)
    _hip.depthwise_conv2d_same(g, out, wt)
    if not np.array_equal(out.numpy(), C.depthwise_conv2d_same(g.numpy(), wt.numpy())):
        bad += 1
        print('MISMATCH', k, planes, h, w)
print('done, mismatches:', bad)
