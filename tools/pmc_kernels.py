#!/usr/bin/env python
"""Mean counter values per kernel of a rocprofv3 --pmc run (csv output) -- for the TIM / DIM kernels' VALU / LDS counters.
    python tools/pmc_kernels.py <rocprofv3 output dir> [substring of the kernel name ...]"""
import collections
import csv
import glob
import os
import sys

keys = sys.argv[2:] or ["dwconv", "dim_"]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if any(k in r["Kernel_Name"] for k in keys):
                rows[r["Kernel_Name"].split("(")[0][-70:] + " grid " + r.get("Grid_Size", "")][r["Counter_Name"]].append(
                    float(r["Counter_Value"]))
for name, counters in rows.items():
    print(name, {c: round(sum(v) / len(v)) for c, v in counters.items()})
