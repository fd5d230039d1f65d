#!/usr/bin/env python
"""The fused update as it runs IN the loop, without the loop: stem kernel (writes gy = dx and the |dx / std| sums, 0.7 ms of
matrix work that also streams 400 MB through the caches) -> ta_mi_update_std, operands rotating over three sets and a 600 MB
copy between iterations (the convolutions' traffic: momentum and delta are never cache-resident in the real loop).  Prints the
update's duration by the dispatch clock next to the stand-alone figure (same launches without the stem kernel in front)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

_hip.load()
N = int(os.environ.get("TA_PROBE_N", "125"))
REPS = 12
gen = torch.Generator().manual_seed(3)
w = (torch.randn(64, 3, 7, 7, generator=gen) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
w2 = _hip.stem7s2_prepare(w)
std = torch.tensor([0.229, 0.224, 0.225], device="cuda")
dy = torch.randn(N, 64, 112, 112, device="cuda").contiguous(memory_format=torch.channels_last) * 1e-3
sets = []
for _ in range(3):
    xb = (torch.randint(0, 256, (N, 3, 224, 224), device="cuda", dtype=torch.uint8).double() / 255).float().contiguous()
    sets.append((torch.randn(N, 3, 224, 224, device="cuda"), torch.zeros(N, 3, 224, 224, device="cuda"), xb, _hip.u8_source_probe(xb)))
big_a = torch.empty(150 * 1024 * 1024, device="cuda")           # 600 MB
big_b = torch.empty_like(big_a)
e = 3 * 224 * 224


tiny = torch.zeros(64, device="cuda")
tim_w = torch.rand(15, 15, device="cuda")
tim_w = (tim_w / tim_w.sum()).contiguous()
tim_in, tim_out = torch.randn(N, 3, 224, 224, device="cuda"), torch.empty(N, 3, 224, 224, device="cuda")


def run(tag, with_stem, flush, bridge=None):
    _hip.timing_begin(REPS + 4)
    for i in range(REPS):
        m, d, x, src = sets[i % 3]
        if flush:
            big_b.copy_(big_a)
        if with_stem:
            gy = _hip.stem7s2_input_grad(dy, w2, torch.empty(N, 3, 224, 224, device="cuda"), std=std)
        else:
            gy = torch.empty(N, 3, 224, 224, device="cuda").copy_(sets[(i + 1) % 3][0]).mul_(1e-4)
            _hip.abs_sum_partials_std(gy, std)
        if bridge == "noop":
            tiny.add_(1.0)                                   # a ~2 us kernel between the producer and the update
        elif bridge == "copy":
            big_b.copy_(big_a)                               # ~200 us of pure memory traffic right before the update
        elif bridge == "tim":
            saved = getattr(gy, "_ta_partials", None)
            _hip.depthwise_conv2d_same(tim_in, tim_out, tim_w)   # ~110 us of VALU work (no matrix cores) right before the update
        elif bridge == "k1":
            saved = gy.__dict__.pop("_ta_partials", None)    # drop the stem's sums: the update runs its sum-only pass first
        _hip.mi_update(gy, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, data_u8=src, std=std)
    torch.cuda.synchronize()
    ms = _hip.timing_end()[3:]
    us = 1e3 * sum(ms) / len(ms)
    print("%-70s %.2f us  (min %.2f, max %.2f)  = %.3f of 8 TB/s at the 24-B contract, %.3f executed"
          % (tag, us, 1e3 * min(ms), 1e3 * max(ms), 24 * e * N / us / 1e3 / 8000, 21 * e * N / us / 1e3 / 8000), flush=True)


print("N = %d, knobs: TA_K2_NT=%s" % (N, os.environ.get("TA_K2_NT", "auto")))
run("stand-alone (K1-std pass in front, operands rotating)", False, False)
run("stand-alone, 600 MB copy between iterations", False, True)
run("after the stem kernel (as in the loop), 600 MB copy between iterations", True, True)
run("after the stem kernel, no copy", True, False)
run("stem kernel -> 2 us kernel -> update", True, False, "noop")
run("stem kernel -> 600 MB copy -> update", True, False, "copy")
run("stem kernel -> TIM convolution (VALU) -> update", True, False, "tim")
run("stem kernel -> sum-only pass + update (timed together)", True, False, "k1")
run("600 MB copy -> update (no stem kernel)", False, False, "copy")
