#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace pass that survives the deletion of the (large) per-dispatch trace.

    python scripts/trace_summary.py <dir with *kernel_trace.csv> <out.json> [<out.csv>]

For every kernel: all dispatches and the WORKING dispatches -- a PCG solve under hipGraph replay keeps up to two replays
in flight past convergence, whose kernels leave at the stop flag after 1.5-3 us; a dispatch counts as working when it
took at least max(3 us, 0.3 x the kernel's 90th percentile) (kernels that never take 3 us keep all their dispatches) --
with median / 10th / 90th percentile / mean of the working ones.  `bytes / median / 8e12` of the dominant kernel is what
bench.py's roofline reports (isolated launches: the *_plain pass; inside a replayed solve: the *_graph pass).
Kernel names are normalised as in scripts/pmc_summary.py, so they match bench.py's kernel table.
"""
import collections
import csv
import glob
import json
import re
import sys

import numpy as np


import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))


def main():
    d, out_json = sys.argv[1], sys.argv[2]
    out_csv = sys.argv[3] if len(sys.argv) > 3 else None
    dur = collections.defaultdict(list)
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("void ", "").replace("sgo::(anonymous namespace)::", "")
                name = re.sub(r"\(.*", "", name)
                dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    res = {}
    for name, v in dur.items():
        a = np.array(v)
        thr = max(3.0, 0.3 * float(np.percentile(a, 90)))
        w = a[a >= thr]
        if w.size == 0:
            w, thr = a, 0.0
        q = np.percentile(w, [10, 50, 90])
        res[name] = {"dispatches": int(a.size), "working": int(w.size), "threshold_us": round(thr, 2),
                     "median_us": round(float(q[1]), 3), "p10_us": round(float(q[0]), 3), "p90_us": round(float(q[2]), 3),
                     "mean_working_us": round(float(w.mean()), 3), "mean_all_us": round(float(a.mean()), 3),
                     "sum_working_ms": round(float(w.sum()) * 1e-3, 3), "min_us": round(float(a.min()), 3), "max_us": round(float(a.max()), 3)}
    order = sorted(res, key=lambda k: -res[k]["sum_working_ms"])
    json.dump({"unit": "microseconds per dispatch", "rule": "working = duration >= max(3 us, 0.3 x p90) (early-exit launches past "
               "convergence dropped); kernels that never reach 3 us keep all dispatches", "trace_files": len(files),
               "kernel_source_sha16": __import__("pmc_summary").kernel_source_sha16(),
               "kernels": {k: res[k] for k in order}}, open(out_json, "w"), indent=1)
    if out_csv:
        with open(out_csv, "w") as fh:
            cols = ["dispatches", "working", "median_us", "p10_us", "p90_us", "mean_working_us", "mean_all_us", "sum_working_ms", "min_us", "max_us"]
            fh.write("kernel," + ",".join(cols) + "\n")
            for k in order:
                fh.write('"' + k + '",' + ",".join(str(res[k][c]) for c in cols) + "\n")


if __name__ == "__main__":
    main()
