#!/usr/bin/env python
"""The stem's input-gradient kernel alone (csrc/stem.hip): event timing at the bench's batch (125) and at 160, with and without
the |dx / std| sums, plus its error against an fp64 evaluation on a small case."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

_hip.load()
gen = torch.Generator().manual_seed(7)
w = (torch.randn(64, 3, 7, 7, generator=gen) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
w2 = _hip.stem7s2_prepare(w)
std = torch.tensor([0.229, 0.224, 0.225], device="cuda")
for n in (125, 160):
    dys = [torch.randn(n, 64, 112, 112, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(2)]
    dx = torch.empty(n, 3, 224, 224, device="cuda")
    for tag, kw in (("plain", {}), ("with the |dx / std| sums", dict(std=std))):
        for i in range(3):
            _hip.stem7s2_input_grad(dys[i % 2], w2, dx, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for i in range(10):
            _hip.stem7s2_input_grad(dys[i % 2], w2, dx, **kw)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 10
        flop = 2.0 * n * 3 * 224 * 224 * 64 * 49 / 4
        print("n=%d stem7s2_input_grad %s: %.1f us  (%.1f TFLOP/s useful)" % (n, tag, us, flop / us / 1e6))
    del dys, dx
dy = torch.randn(2, 64, 40, 40, generator=gen)
truth = torch.ops.aten.convolution_backward(dy.double(), torch.empty(2, 3, 80, 80).double(), w.cpu().double().contiguous(), None, [2, 2], [3, 3],
                                            [1, 1], False, [0, 0], 1, [True, False, False])[0]
got = _hip.stem7s2_input_grad(dy.cuda().contiguous(memory_format=torch.channels_last), w2, torch.empty(2, 3, 80, 80, device="cuda")).cpu().double()
print("max error / max|dx| vs fp64: %.2e" % (float((got - truth).abs().max()) / float(truth.abs().max())))
