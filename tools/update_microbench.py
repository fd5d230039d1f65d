#!/usr/bin/env python
"""Stand-alone driver of the fused update kernels for profiling (rocprofv3 --pmc / --kernel-trace).
Runs, for N in {32, 250}: a known-size device copy (calibration of the byte counters), the two-launch update and
the single-launch update, rotating over 4 operand sets so the 256 MiB Infinity Cache cannot serve them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

REPS = int(os.environ.get("TA_MICRO_REPS", "8"))


def main():
    _hip.load()
    for n in (32, 250):
        sets = []
        for _ in range(4):
            g = torch.randn(n, 3, 224, 224, device="cuda") * 1e-4
            sets.append((g, torch.randn_like(g), torch.zeros_like(g), torch.rand_like(g)))
        dst = torch.empty_like(sets[0][0])
        for i in range(REPS):
            dst.copy_(sets[i % 4][1])                       # calibration: reads 4 B/elem, writes 4 B/elem
        for single in (False, True):
            for i in range(REPS):
                g, m, d, x = sets[i % 4]
                _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, single_launch=single)
        torch.cuda.synchronize()
        del sets, dst
    print("microbench done")


if __name__ == "__main__":
    main()
