"""The fused update alone, timed by HIP events on its own dispatch packet (ta_timing_begin / ta_timing_end), at the
launch shapes of the bench lines: N = 32 (DTS / ENS / reference-literal loops) and N = 125 (headline).  The |g| tile sums
are handed over by a producer (here K1, which like the loop's producers leaves g in the Infinity Cache), the kernel
writes x + delta for the next iteration: 28 B/element.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transferattack_amd import _hip  # noqa: E402


def main():
    dev = "cuda"
    out = {}
    for n in (32, 125):
        shape = (n, 3, 224, 224)
        sets = max(2, int(1.2e9 // (6 * 4 * n * 150528)))          # rotate operands over > 1 GB: nothing but g stays cached
        ops = [dict(g=torch.randn(shape, device=dev) * 1e-4, m=torch.randn(shape, device=dev),
                    d=torch.zeros(shape, device=dev), x=torch.rand(shape, device=dev), xa=torch.empty(shape, device=dev))
               for _ in range(sets)]
        reps = 40
        for timed in (False, True):
            if timed:
                _hip.timing_begin(reps)
            for r in range(reps):
                o = ops[r % sets]
                _hip.abs_sum_partials(o["g"])
                _hip.mi_update(o["g"], o["m"], o["m"], o["d"], o["x"], 1.0, 1.6 / 255, 16 / 255, x_adv=o["xa"])
            torch.cuda.synchronize()
        ms = _hip.timing_end()
        us = sorted(1e3 * t for t in ms)
        mean = sum(us) / len(us)
        nbytes = 28 * n * 150528
        out["n%d" % n] = {"launches": len(us), "mean_us": round(mean, 2), "median_us": round(us[len(us) // 2], 2),
                          "min_us": round(us[0], 2), "TBps": round(nbytes / mean / 1e6, 3),
                          "frac_of_8TBps": round(nbytes / mean / 1e6 / 8.0, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
