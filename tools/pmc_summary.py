#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/update_microbench.py: bytes per launch per
kernel, calibrated on the known-size copy kernel of the same run (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
reports half of a wide coalesced read -- the copy kernel measures that factor instead of assuming it)."""
import csv
import glob
import os
import sys
from collections import defaultdict

E = 3 * 224 * 224


def load(out, counter):
    rows = defaultdict(list)
    for path in glob.glob(os.path.join(out, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                grid = int(r.get("Grid_Size", "0") or 0)
                rows[(r["Kernel_Name"], grid)].append(float(r["Counter_Value"]))
    return rows


def main():
    out = sys.argv[1]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = load(out, counter)
        print("== %s (KiB per launch, mean over launches)" % counter)
        for (name, grid), vals in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            short = name.split("(")[0][-70:]
            mean_kib = sum(vals) / len(vals)
            n_img = None
            for n in (32, 250):
                if grid in (49 * n * 256,):
                    n_img = n
            per_elem = "  = %.2f B/elem" % (mean_kib * 1024 / (E * n_img)) if n_img else ""
            print("%-72s grid %-10d launches %-4d mean %.1f KiB%s" % (short, grid, len(vals), mean_kib, per_elem))


if __name__ == "__main__":
    main()
