#!/usr/bin/env python
"""Can the stem's forward chain (MIOpen zero fill + 7x7/2 convolution on 3 channels + ta_bias_act) become ONE composable_kernel
convolution with bias + ReLU as epilogue?  CK's vector loads run along C, so the image gets a fourth, all-zero channel (and the
filter a fourth all-zero input plane): times every TA_CK_FWD_BIAS_RELU configuration on [n, 4, 224, 224] -> 64 against the
two-kernel form on 3 channels, and the cost of the padding pass.   python tools/ck_stem_probe.py [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _ck, _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125
torch.backends.cudnn.benchmark = True
_hip.load()
lib = _ck.load()
CL = torch.channels_last
x3 = torch.randn(n, 3, 224, 224, device="cuda").contiguous(memory_format=CL)
w3 = (torch.randn(64, 3, 7, 7, device="cuda") * 0.05).contiguous(memory_format=CL)
bias = torch.randn(64, device="cuda")


def two_kernels():
    y = F.conv2d(x3, w3, None, 2, 3)
    _hip.bias_act_(y, bias)
    return y


def pad4():
    x4 = torch.zeros(n, 4, 224, 224, device="cuda").contiguous(memory_format=CL)
    x4[:, :3] = x3
    return x4


ref = two_kernels()
print("two kernels (MIOpen 3-channel conv + bias_act): %.1f us" % (_ck._time(two_kernels) * 1e3))
print("zero-padding pass (torch): %.1f us" % (_ck._time(pad4) * 1e3))
x4 = pad4()
w4 = torch.zeros(64, 4, 7, 7, device="cuda")
w4[:, :3] = w3
w4 = w4.permute(0, 2, 3, 1).contiguous()
geom = (n, 4, 224, 224, 64, 7, 2, 3)
out = torch.empty_like(ref)
best = (float("inf"), None)
for idx in range(lib.ta_ck_instances(_ck.FWD_BIAS_RELU, 7, 2, 3)):
    if _ck.conv(_ck.FWD_BIAS_RELU, idx, x4, w4, bias, None, None, out, geom) != 0:
        continue
    ms = _ck._time(lambda: _ck.conv(_ck.FWD_BIAS_RELU, idx, x4, w4, bias, None, None, out, geom))
    err = float((out - ref).abs().max() / ref.abs().max())
    name = lib.ta_ck_instance_name(_ck.FWD_BIAS_RELU, 7, 2, 3, idx).decode()
    print("  %-110s %.1f us  (max err %.1e)" % (name[:110], ms * 1e3, err))
    best = min(best, (ms, name))
print("best fused: %.1f us  %s" % (best[0] * 1e3, best[1]))
