#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/update_microbench.py: bytes per launch per
kernel, calibrated on the known-size copy kernel of the same run (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
reports half of a wide coalesced read -- the copy kernel measures that factor instead of assuming it).
    python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> [N] [--json profiles/pmc_update_kernel.json]"""
import csv
import glob
import json
import os
import re
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
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 125
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = load(out, counter)
        print("== %s (KiB per launch, mean over launches; N = %d)" % (counter, n))
        for name, vals in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            short = re.sub(r"\(.*", "", name)[-90:]
            mean_kib = sum(vals) / len(vals)
            b_per_elem = mean_kib * 1024 / (E * n)
            per.setdefault(short, {})[counter] = b_per_elem
            print("%-92s launches %-4d mean %.1f KiB = %.3f B/elem" % (short, len(vals), mean_kib, b_per_elem))
    copy = [k for k in per if "elementwise" in k or "copy" in k.lower()]
    factor = None
    if "--factor" in sys.argv:                      # a calibration taken elsewhere (say where): this run's copy launches are not
        factor = float(sys.argv[sys.argv.index("--factor") + 1])       # all of one size (e.g. host-to-device staging copies)
        print("fetch correction x%.4f given on the command line" % factor)
    elif any("abs_sum_partials_kernel<4, false, false, false>" in k for k in per):
        # the calibration of choice: K1 is a pure, fully coalesced read of exactly 4 B/element at this very N (the device copies of
        # a run mix shapes -- staging copies, index tensors -- and gave x7.9 in r6b where this kernel gives x1.999)
        k1 = next(k for k in per if "abs_sum_partials_kernel<4, false, false, false>" in k)
        factor = 4.0 / per[k1]["FETCH_SIZE"]
        print("calibration on %s: reported fetch %.3f B/elem for a pure read of 4 B/elem -> correction x%.3f" % (k1[-60:], per[k1]["FETCH_SIZE"], factor))
    elif copy:
        c = per[copy[0]]
        if c.get("FETCH_SIZE"):
            factor = 4.0 / c["FETCH_SIZE"]                # the copy reads 4 B/elem
            print("calibration on %s: reported fetch %.3f B/elem for 4 B/elem read -> correction x%.3f; write %.3f for 4"
                  % (copy[0][-40:], c["FETCH_SIZE"], factor, c.get("WRITE_SIZE", float("nan"))))
    if "--json" in sys.argv and factor:
        dst = sys.argv[sys.argv.index("--json") + 1]
        doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/update_microbench.py, "
                         "N = %d; summary by tools/pmc_summary.py" % n,
               "fetch_correction": round(factor, 4),
               "fetch_correction_source": "command line (--factor)" if "--factor" in sys.argv else (
                   "this run's abs_sum_partials_kernel launches: a pure read of exactly 4 B/element" if any(
                       "abs_sum_partials_kernel<4, false, false, false>" in k for k in per) else "the known-size device copy of the same run"),
               "kernels": {}}
        for name, c in per.items():
            if "ta::" not in name and "ta2" not in name:
                continue
            doc["kernels"][name] = {"fetch_B_per_elem_reported": round(c.get("FETCH_SIZE", 0.0), 3),
                                    "fetch_B_per_elem_corrected": round(c.get("FETCH_SIZE", 0.0) * factor, 3),
                                    "write_B_per_elem": round(c.get("WRITE_SIZE", 0.0), 3)}
        json.dump(doc, open(dst, "w"), indent=1)
        print("wrote", dst)


if __name__ == "__main__":
    main()
