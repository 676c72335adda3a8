#!/usr/bin/env python3
"""optimize(iters) of a named config, `repeat` times in one process (the later passes at busy-chip clocks): solver description,
set-up time, PCG counts, per-iteration times.  Usage: python scripts/cfg_probe.py CONFIG [iters] [repeat] [key=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = synth.config(cfg, **dict(a.split("=", 1) for a in sys.argv[4:]))
with capi.Optimizer(0) as o:
    for rep in range(repeat):
        t = time.perf_counter()
        o.set_graph(*g.arrays())
        ts = time.perf_counter() - t
        t = time.perf_counter()
        d, st = o.optimize(iters)
        to = time.perf_counter() - t
        print(f"pass {rep}: set_graph {1e3 * ts:.1f} ms, optimize({iters}) {1e3 * to:.1f} ms = {g.E * iters / to / 1e6:.1f} M edge-Jacobians/s, done {d}"
              + ("" if d == iters else " error: " + o.last_error()))
        print("  pcg", st["pcg_iters"], "sum", sum(st["pcg_iters"]), "median ms per GN iteration", round(1e3 * float(np.median(st["seconds"])), 2))
        print("  ms per GN iteration", [round(1e3 * x, 1) for x in st["seconds"]])
        print("  of which linearise + hierarchy refresh", [round(1e3 * x, 1) for x in st["seconds_linearize"]], "solve", [round(1e3 * x, 1) for x in st["seconds_solve"]])
    print(o.solver_description())
