#!/usr/bin/env python3
"""Round 6: what a rebuild inside sgo_optimize_gn loses when it KEEPS the aggregates of the hierarchy it replaces (and re-makes
only the filter's mask, the patterns and the values) -- the question behind a device-side rebuild.  optimize(20) from BASELINE.md's
literal dead-reckoned start, SGO_AMG_KEEP_AGG=0 against 1, second pass of each (the first warms the clocks).
Usage: python scripts/keep_agg_probe.py [config ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

for cfg in (sys.argv[1:] or ["C4", "C2"]):
    g = synth.config(cfg, init="odom")
    for keep in ("0", "1", "0", "1"):
        os.environ["SGO_AMG_KEEP_AGG"] = keep
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            t = time.perf_counter()
            d, st = o.optimize(20)
            dt = time.perf_counter() - t
            print(f"{cfg} keep_agg={keep}: done {d}, call {1e3 * dt:.1f} ms, median GN iteration {1e3 * float(np.median(st['seconds'])):.2f} ms, "
                  f"sum of iterations {1e3 * float(np.sum(st['seconds'])):.1f} ms, pcg {st['pcg_iters']}, converged {all(st['pcg_converged'][:d])}", flush=True)
            print("   ", o.solver_description()[:400], flush=True)
