#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic.

Usage (on the GPU box, one pass per counter as MI355X_MICROARCH.md prescribes):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE -- python3 bench.py ...
    python scripts/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/rNN_pmc_traffic.json

Corrections (MI355X_MICROARCH.md section HBM): counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the fetched bytes (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact.
The factor was re-checked on this repo's own access patterns with known byte counts:
k_update_p (reads 2 vectors, writes 1) and k_update_xr (reads 4 vectors + dinv) on C4.
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    name = row["Kernel_Name"].replace("void ", "").replace("sgo::(anonymous namespace)::", "")
                    name = re.sub(r"\(.*", "", name)
                    out[name].append(float(row["Counter_Value"]))
    return out



def kernel_source_sha16():
    """sha256 (first 16 hex digits) of the sources the solve's kernels are compiled from (sgo_kernels.hip, sgo_device.h): recorded
    with every summary so that bench.py can say whether the profile it quotes was taken on the kernels it runs."""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("sgo_kernels.hip", "sgo_device.h"):
        h.update(open(os.path.join(root, "sparse_gslam_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(W)):
        f, w = F.get(k, []), W.get(k, [])
        n = max(len(f), len(w))
        fetch = 2.0 * 1024.0 * (sum(f) / len(f)) if f else 0.0
        write = 1024.0 * (sum(w) / len(w)) if w else 0.0
        res[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                  "hbm_bytes_per_launch": fetch + write}
    json.dump({"unit": "bytes per launch (mean over launches)", "corrections": "FETCH_SIZE x2 x1024, WRITE_SIZE x1024",
               "kernel_source_sha16": kernel_source_sha16(), "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
