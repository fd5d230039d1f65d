"""Time ta_dim_fwd / ta_dim_bwd on the GPU for the variant the environment selects (TA_DIM_FWD_VARIANT,
TA_DIM_BWD_VARIANT); the knobs are read once per process, so run once per setting (tools/gpu_check.sh dimvariants)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402

_hip.load()
tag = "fwd_variant=%s bwd_variant=%s" % (os.environ.get("TA_DIM_FWD_VARIANT", "0"), os.environ.get("TA_DIM_BWD_VARIANT", "0"))
for n in (32, 160):
    xs = [torch.rand(n, 3, 224, 224, device="cuda") for _ in range(3)]        # rotate inputs: no warm-cache flattery
    y = torch.empty_like(xs[0])
    for name, fn in (("dim_fwd", _hip.dim_fwd), ("dim_bwd", _hip.dim_bwd)):
        for i in range(4):
            fn(xs[i % 3], y, 246, 237, 3, 5)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for i in range(20):
            fn(xs[i % 3], y, 246, 237, 3, 5)
        end.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(end) * 1e3 / 20
        print("%s n=%d %s %.2f us  (%.2f TB/s at 8 B/element)" % (tag, n, name, us, n * 3 * 224 * 224 * 8 / us / 1e6))
