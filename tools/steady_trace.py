#!/usr/bin/env python
"""Post-process a rocprofv3 --kernel-trace CSV of bench.py ON THE GPU BOX (the trace itself is too large to bring back):

  1. per-kernel totals over the STEADY-STATE part only (everything after the first fused update of the timed region's
     second step; MIOpen find-mode trial kernels, if any survive the warmed find-db, fall before it), top N by time, with
     the share of the step they account for and a coarse class (conv / gemm / norm+elementwise / layout / ta:: / other);
  2. the dispatch sequence of ONE attack iteration (between two consecutive mi_update launches): name, duration, grid,
     workgroup -- which shows what the surrogate's forward / backward is made of, in order.

usage: steady_trace.py <dir with *kernel_trace.csv> <out.json> [top_n]
"""
import csv
import glob
import json
import os
import sys


def classify(name):
    n = name.lower()
    if n.startswith("ta::") or "ta::" in n:
        return "ta"
    if "conv" in n or "igemm" in n or "winograd" in n or "sp3" in n or "gridwise" in n and "conv" in n:
        return "conv"
    if "cijk" in n or "gemm" in n or "_mt" in n and "mi" in n:
        return "gemm"
    if "transpose" in n or "batched_transpose" in n or "nchw" in n or "nhwc" in n or "tensor_reorder" in n:
        return "layout"
    if "batchnorm" in n or "batch_norm" in n or "elementwise" in n or "vectorized" in n or "reduce" in n or "softmax" in n \
            or "pool" in n or "threshold" in n or "relu" in n:
        return "elementwise"
    return "other"


def main():
    src, out = sys.argv[1], sys.argv[2]
    top_n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    paths = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for path in paths:
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             [int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0)],
                             [int(r.get("Workgroup_Size_X", 0) or 0), int(r.get("Workgroup_Size_Y", 0) or 0),
                              int(r.get("Workgroup_Size_Z", 0) or 0)],
                             int(r.get("LDS_Block_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0)))
    rows.sort()
    upd = [i for i, r in enumerate(rows) if "mi_update_kernel" in r[2]]
    if len(upd) < 40:
        raise SystemExit("only %d fused updates in the trace" % len(upd))
    # steady state: from the update that ends iteration 10 of the first TIMED step onwards is conservative; simpler and
    # robust: the second half of the updates
    first = upd[len(upd) // 2]
    steady = rows[first + 1:upd[-1] + 1]
    wall = rows[upd[-1]][1] - rows[first][1]
    busy = sum(e - s for s, e, *_ in steady)
    per = {}
    for s, e, name, grid, wg, lds, vgpr in steady:
        d = per.setdefault(name, [0, 0, grid, wg, lds, vgpr])
        d[0] += e - s
        d[1] += 1
    iters = len(upd) - len(upd) // 2 - 1 + 1 - 1 + 1          # updates inside (first, last]
    iters = len([i for i in upd if first < i <= upd[-1]])
    top = sorted(per.items(), key=lambda kv: -kv[1][0])[:top_n]
    classes = {}
    for name, d in per.items():
        c = classes.setdefault(classify(name), [0, 0])
        c[0] += d[0]
        c[1] += d[1]
    # one iteration: between two consecutive updates in the middle of the steady part
    mid = [i for i in upd if i > first][len([i for i in upd if i > first]) // 2]
    prev = max(i for i in upd if i < mid)
    one = [{"name": n[:110], "us": round((e - s) / 1e3, 2), "grid": g, "wg": w, "lds": l, "vgpr": v,
            "gap_us": None} for s, e, n, g, w, l, v in rows[prev + 1:mid + 1]]
    seq = rows[prev + 1:mid + 1]
    for k in range(1, len(seq)):
        one[k]["gap_us"] = round((seq[k][0] - seq[k - 1][1]) / 1e3, 2)
    result = {
        "iterations_in_steady_part": iters,
        "wall_us_per_iteration": round(wall / 1e3 / iters, 1),
        "kernel_busy_us_per_iteration": round(busy / 1e3 / iters, 1),
        "gpu_idle_share": round(1 - busy / wall, 4),
        "launches_per_iteration": round(len(steady) / iters, 1),
        "classes": {k: {"us_per_iteration": round(v[0] / 1e3 / iters, 1), "launches_per_iteration": round(v[1] / iters, 1),
                        "share_of_busy": round(v[0] / busy, 4)} for k, v in sorted(classes.items(), key=lambda kv: -kv[1][0])},
        "top": [{"name": n[:160], "class": classify(n), "us_per_iteration": round(d[0] / 1e3 / iters, 2),
                 "calls_per_iteration": round(d[1] / iters, 2), "mean_us": round(d[0] / 1e3 / d[1], 2),
                 "share_of_busy": round(d[0] / busy, 4), "grid": d[2], "wg": d[3], "lds": d[4], "vgpr": d[5]} for n, d in top],
        "one_iteration": one,
    }
    with open(out, "w") as fh:
        json.dump(result, fh, indent=1)
    print(json.dumps({k: result[k] for k in ("iterations_in_steady_part", "wall_us_per_iteration",
                                               "kernel_busy_us_per_iteration", "gpu_idle_share", "launches_per_iteration",
                                               "classes")}))
    for t in result["top"][:12]:
        print("%8.1f us/it %6.2f calls %5.1f%%  [%s] %s" % (t["us_per_iteration"], t["calls_per_iteration"],
                                                          100 * t["share_of_busy"], t["class"], t["name"][:100]))


if __name__ == "__main__":
    main()
