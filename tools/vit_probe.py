#!/usr/bin/env python
"""One ViT-B/16 surrogate evaluation (forward + input gradient, fp32, batch 160 = 5 stacked VMI neighbours of 32 images) under
the attention / GEMM back ends PyTorch-ROCm offers: which one should configs[3] run on?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import backbones  # noqa: E402

os.environ.setdefault("TA_ALLOW_RANDOM_INIT", "1")
net = backbones.create("vit_base_patch16_224", seed=0, verbose=False).cuda().eval()
for p in net.parameters():
    p.requires_grad_(False)
n = int(os.environ.get("TA_VIT_N", "160"))
x = torch.randn(n, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (n,), device="cuda")


def evaluate():
    xin = x.clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(net(xin), y)
    return torch.autograd.grad(loss, xin)[0]


def timed(tag, ctx=None):
    import contextlib
    with (ctx if ctx is not None else contextlib.nullcontext()):
        try:
            for _ in range(2):
                evaluate()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                g = evaluate()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            print("%-44s %.1f ms per evaluation of %d images (%.1f TFLOP/s at 70 GFLOP/image)   |g| %.4e" % (
                tag, ms, n, 70e9 * n / ms / 1e9, float(g.abs().mean())), flush=True)
        except Exception as exc:  # noqa: BLE001
            print("%-44s failed: %s" % (tag, repr(exc)[:120]), flush=True)


from torch.nn.attention import SDPBackend, sdpa_kernel  # noqa: E402
print("preferred BLAS library:", torch.backends.cuda.preferred_blas_library())
timed("default")
timed("attention: math", sdpa_kernel([SDPBackend.MATH]))
timed("attention: memory-efficient", sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]))
timed("attention: flash", sdpa_kernel([SDPBackend.FLASH_ATTENTION]))
for lib in ("hipblaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        timed("GEMM library: %s" % lib)
    except Exception as exc:  # noqa: BLE001
        print("GEMM library %s: %s" % (lib, repr(exc)[:100]))
