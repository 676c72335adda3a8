#!/usr/bin/env python3
"""Round 6: the count rules' cost constant (what a set-up is worth in PCG iterations) now that a rebuild inside sgo_optimize_gn costs
16-18 ms instead of 50: optimize(20) from BASELINE.md's dead-reckoned start, second pass of each value.
Usage: python scripts/rebuild_cost_sweep.py [config ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

for spec in (sys.argv[1:] or ["C4", "C2"]):
    g = synth.config(spec, init="odom")
    for cost in (30, 60, 100, 150, 300):
        os.environ["SGO_AMG_REBUILD_COST"] = str(cost)
        with capi.Optimizer(0) as o:
            res = []
            for _ in range(2):
                o.set_graph(*g.arrays())
                t = time.perf_counter()
                d, st = o.optimize(20)
                res.append(time.perf_counter() - t)
            print(f"{spec} cost {cost:4d}: done {d}, call {1e3 * res[-1]:7.1f} ms (first pass {1e3 * res[0]:7.1f}), median GN iteration {1e3 * float(np.median(st['seconds'])):6.2f} ms, "
                  f"pcg sum {int(sum(st['pcg_iters']))} {st['pcg_iters']}", flush=True)
