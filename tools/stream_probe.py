#!/usr/bin/env python
"""Does running two half-batches on two HIP streams (memory-bound elementwise kernels of one overlapping the
convolutions of the other) beat one full batch on one stream?  ResNet-50, folded BN, NHWC, fwd + input-grad bwd."""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import backbones  # noqa: E402
from transferattack_amd.utils import wrap_model  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
model = backbones.create("resnet50", verbose=False)
for p in model.parameters():
    p.requires_grad_(False)
backbones.fold_batchnorm(model)
model = wrap_model(model.to(dev)).to(memory_format=torch.channels_last)
ce = torch.nn.CrossEntropyLoss()
ITERS = 10


def work(batch, stream, out, key):
    with torch.cuda.stream(stream):
        x = torch.rand(batch, 3, 224, 224, device=dev)
        y = torch.randint(0, 1000, (batch,), device=dev)
        delta = torch.zeros_like(x, requires_grad=True)
        for _ in range(ITERS):
            g = torch.autograd.grad(ce(model(x + delta), y), delta)[0]
        stream.synchronize()
    out[key] = float(g.abs().sum())


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def one_stream(total):
    work(total, torch.cuda.current_stream(), {}, 0)


def two_streams(total):
    out = {}
    ts = [threading.Thread(target=work, args=(total // 2, torch.cuda.Stream(), out, i)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


res = {}
for total in (128, 250):
    a = timed(lambda: one_stream(total))
    b = timed(lambda: two_streams(total))
    res["batch%d" % total] = {"one_stream_img_iter_per_s": round(total * ITERS / a, 1),
                             "two_streams_img_iter_per_s": round(total * ITERS / b, 1)}
print(json.dumps(res))
