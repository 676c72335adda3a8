#!/usr/bin/env python3
"""BASELINE.md's literal workload -- C4 from a dead-reckoned start -- : the hierarchy the set-up makes for it and the PCG
iterations of the first Gauss-Newton iterations.  Usage: python scripts/odom_probe.py [config] [iters] [repeat]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = synth.config(cfg, init="odom")
with capi.Optimizer(0) as o:
    for rep in range(repeat):   # (a second pass runs at the clocks of a busy chip: the first one's times are those of a cold start)
        o.set_graph(*g.arrays())
        print(o.solver_description())
        d, st = o.optimize(iters)
        print("done", d, "of", iters, "" if d == iters else "error: " + o.last_error())
        print("pcg", st["pcg_iters"], "relres", [f"{r:.1e}" for r in st["pcg_relres"]], "ms", [round(1e3 * s, 1) for s in st["seconds"]],
              "median ms", round(1e3 * float(np.median(st["seconds"])), 2) if st["seconds"] else None, "robust chi2", [f"{c:.4g}" for c in st["robust_chi2"]])
        print(o.solver_description())
