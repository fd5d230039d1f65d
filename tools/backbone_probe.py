#!/usr/bin/env python
"""GPU probe (not part of the product path): how fast is one surrogate forward + input-gradient backward of
ResNet-50 fp32 on MI355X under different host-side arrangements?  Decides the defaults of Attack / bench.py.

    python tools/backbone_probe.py [--batches 32,125] [--model resnet50]
Variants: memory format (NCHW / channels_last), BatchNorm folded into the convolutions (eval-mode algebra),
hipGraph capture of the whole iteration.  Prints one JSON line per (variant, batch).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import backbones  # noqa: E402
from transferattack_amd.utils import wrap_model  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,125")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda")
    ce = torch.nn.CrossEntropyLoss()
    for fold in (False, True):
        for nhwc in (False, True):
            model = backbones.create(args.model, verbose=False)
            for p in model.parameters():
                p.requires_grad_(False)
            if fold:
                backbones.fold_batchnorm(model)
            model = wrap_model(model.to(dev))
            if nhwc:
                model = model.to(memory_format=torch.channels_last)
            for batch in [int(b) for b in args.batches.split(",")]:
                x = torch.rand(batch, 3, 224, 224, device=dev)
                y = torch.randint(0, 1000, (batch,), device=dev)
                delta = torch.zeros_like(x, requires_grad=True)

                def it():
                    xin = x + delta
                    if nhwc:
                        xin = xin.contiguous(memory_format=torch.channels_last)
                    loss = ce(model(xin), y)
                    return torch.autograd.grad(loss, delta)[0]

                try:
                    ms = timed(it, args.reps)
                    rec = {"model": args.model, "fold_bn": fold, "channels_last": nhwc, "batch": batch, "graph": False,
                           "ms_per_iter": round(ms, 3), "images_iter_per_s": round(batch / ms * 1e3, 1)}
                    print(json.dumps(rec), flush=True)
                    # hipGraph capture of the same iteration
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        for _ in range(3):
                            it()
                    torch.cuda.current_stream().wait_stream(s)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=s):
                        g_static = it()
                    ms = timed(graph.replay, args.reps)
                    rec.update(graph=True, ms_per_iter=round(ms, 3), images_iter_per_s=round(batch / ms * 1e3, 1))
                    print(json.dumps(rec), flush=True)
                    del graph, g_static
                except Exception as exc:  # noqa: BLE001
                    print(json.dumps({"fold_bn": fold, "channels_last": nhwc, "batch": batch, "error": repr(exc)[:300]}),
                          flush=True)
                del x, y, delta
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
