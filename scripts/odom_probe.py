#!/usr/bin/env python3
"""BASELINE.md's literal workload -- C4 from a dead-reckoned start -- : the hierarchy the set-up makes for it and the PCG
iterations of the first Gauss-Newton iterations.  Usage: python scripts/odom_probe.py [config] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = synth.config(cfg, init="odom")
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    print(o.solver_description())
    d, st = o.optimize(iters)
    print("pcg", st["pcg_iters"], "ms", [round(1e3 * s, 1) for s in st["seconds"]], "robust chi2", [f"{c:.4g}" for c in st["robust_chi2"]])
    print(o.solver_description())
