"""ta_bsr_fwd / ta_bsr_bwd alone at the attack's shapes (20 copies, 3 x 3 blocks; batch 16 = the bench line, 32 = the
kernel table), HIP events around batches of launches.  Prints one JSON line."""
import json
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402
from transferattack_amd.transforms import bsr_draw  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / reps


def main():
    out = {}
    copies, nb = 20, 3
    for n in (16, 32):
        shape = (n, 3, 224, 224)
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        plan = torch.from_numpy(np.ascontiguousarray(bsr_draw(shape, nb, copies))).cuda()
        x = torch.rand(shape, device="cuda")
        stack = torch.empty((copies * n, 3, 224, 224), device="cuda")
        gx = torch.empty(shape, device="cuda")
        nbytes = 4 * x.numel() * (copies + 1)
        fwd = timed(lambda: _hip.bsr_fwd(x, plan, stack, copies, nb), 6)
        bwd = timed(lambda: _hip.bsr_bwd(stack, plan, gx, copies, nb), 6)
        out["n%d" % n] = {"fwd_us": round(fwd, 1), "fwd_GBps": round(nbytes / fwd / 1e3, 1),
                          "bwd_us": round(bwd, 1), "bwd_GBps": round(nbytes / bwd / 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
