#!/usr/bin/env python
"""Stand-alone driver of the fused update kernels for profiling (rocprofv3 --pmc / --kernel-trace).
Runs at N = TA_MICRO_N (default 125, the bench's launch shape): a known-size device copy (calibration of the byte
counters), then the shipped update in the three shapes the loop issues -- steady state (24 B/element + 4 for x_adv),
first iteration (no momentum read), and K1 + K2 (no producer-side sums) -- rotating over 4 operand sets so the
256 MiB Infinity Cache cannot serve them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

REPS = int(os.environ.get("TA_MICRO_REPS", "8"))
N = int(os.environ.get("TA_MICRO_N", "125"))


def main():
    _hip.load()
    sets = []
    for _ in range(4):
        g = torch.randn(N, 3, 224, 224, device="cuda") * 1e-4
        sets.append((g, torch.randn_like(g), torch.zeros_like(g), torch.rand_like(g), torch.empty_like(g)))
    dst = torch.empty_like(sets[0][0])
    for i in range(REPS):
        dst.copy_(sets[i % 4][1])                       # calibration: reads 4 B/elem, writes 4 B/elem
    for i in range(REPS):                               # K1 + K2 (24 B/elem + 4 for K1's pass)
        g, m, d, x, xa = sets[i % 4]
        _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255)
    for i in range(REPS):                               # steady state of the loop: sums ready, x_adv written (28 B/elem)
        g, m, d, x, xa = sets[i % 4]
        _hip.abs_sum_partials(g)
        _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, x_adv=xa)
    for i in range(REPS):                               # first iteration: no momentum read (24 B/elem with x_adv)
        g, m, d, x, xa = sets[i % 4]
        _hip.abs_sum_partials(g)
        _hip.mi_update(g, None, m, d, x, 1.0, 1.6 / 255, 16 / 255, x_adv=xa)
    for i in range(REPS):                               # decay 0: no momentum at all (20 B/elem with x_adv)
        g, m, d, x, xa = sets[i % 4]
        _hip.abs_sum_partials(g)
        _hip.mi_update(g, None, None, d, x, 0.0, 1.6 / 255, 16 / 255, x_adv=xa)
    # the byte source (ta_mi_update_u8): byte-valued images, 1 B instead of 4 B per element of x
    bsets = []
    for g, m, d, x, xa in sets:
        # byte-valued images built on the device WITHOUT torch's fp32 division (by a scalar it multiplies by the reciprocal,
        # not the IEEE quotient for 126 of the 256 bytes: the probe would -- rightly -- say "not byte-valued", r4b's first PMC
        # pass) and without host-to-device copies (they would join the calibration copy's kernel name, r4c's second): the
        # double-precision product rounds to the correctly rounded fp32 quotient for every byte
        xb = (torch.randint(0, 256, x.shape, device="cuda", dtype=torch.uint8).double() / 255).float().contiguous()
        bsets.append((g, m, d, xb, xa, _hip.u8_source_probe(xb)))
    for i in range(REPS):                               # steady state with the byte source (25 B/element executed)
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.abs_sum_partials(g)
        _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, x_adv=xa, data_u8=src)
    for i in range(REPS):                               # last iteration with the byte source (21 B/element executed)
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.abs_sum_partials(g)
        _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, data_u8=src)
    for i in range(REPS):                               # first iteration with the byte source
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.abs_sum_partials(g)
        _hip.mi_update(g, None, m, d, x, 1.0, 1.6 / 255, 16 / 255, x_adv=xa, data_u8=src)
    # round 5: the std form (the gradient operand is gy, divided by std[c] inline; no x_adv store) and the forward end
    std = torch.tensor([0.229, 0.224, 0.225], device="cuda")
    mean = torch.tensor([0.485, 0.456, 0.406], device="cuda")
    for i in range(REPS):                               # steady state (21 B/element executed)
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.abs_sum_partials_std(g, std)
        _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, data_u8=src, std=std)
    for i in range(REPS):                               # first iteration (17 B/element executed)
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.abs_sum_partials_std(g, std)
        _hip.mi_update(g, None, m, d, x, 1.0, 1.6 / 255, 16 / 255, data_u8=src, std=std)
    for i in range(REPS):                               # decay 0 (13 B/element executed)
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.abs_sum_partials_std(g, std)
        _hip.mi_update(g, None, None, d, x, 0.0, 1.6 / 255, 16 / 255, data_u8=src, std=std)
    for i in range(REPS):                               # ta_normalize_adv_fwd with the byte source (5 B in, 4 B out)
        g, m, d, x, xa, src = bsets[i % 4]
        _hip.normalize_adv_fwd(x, d, xa, mean, std, data_u8=src)
    torch.cuda.synchronize()
    print("byte source taken by the kernels: %s" % all(int(b[5][1].item()) == 0 for b in bsets))
    print("microbench done N=%d" % N)


if __name__ == "__main__":
    main()
