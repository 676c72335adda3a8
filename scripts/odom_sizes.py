#!/usr/bin/env python3
"""BASELINE.md's literal dead-reckoned start over sizes, densities and seeds: optimize(20) must complete (every solve
converged) whatever the hierarchy does on the way.  Usage: python scripts/odom_sizes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = [(5000, 20000, 1), (20000, 100000, 2), (20000, 200000, 3), (50000, 250000, 4), (50000, 500000, 5), (100000, 600000, 6),
         (200000, 2000000, 7), (100000, 1000000, 8), (30000, 90000, 9), (150000, 900000, 10)]
bad = 0
with capi.Optimizer(0) as o:
    for V, E, seed in cases:
        for kw in (dict(), dict(info_mode="full", phi=10.0)):
            g = synth.manhattan(V, E, seed=seed, init="odom", **kw)
            t = time.perf_counter()
            o.set_graph(*g.arrays())
            ts = time.perf_counter() - t
            desc = o.solver_description()
            t = time.perf_counter()
            d, st = o.optimize(20)
            to = time.perf_counter() - t
            ok = d == 20 and all(st["pcg_converged"][:20])
            bad += not ok
            print(f"V={V:7d} E={E:8d} seed={seed:2d} {'full phi=10' if kw else 'diag phi=1 ':11s} {'filtered' if ' filtered' in desc else 'other   '} "
                  f"done={d:2d} set_graph {1e3 * ts:6.1f} ms optimize {1e3 * to:7.1f} ms = {20 * E / to / 1e6:6.1f} M/s median {1e3 * float(np.median(st['seconds'])) if st['seconds'] else 0:6.2f} ms "
                  f"pcg min/median/max {min(st['pcg_iters'][:max(d,1)])}/{int(np.median(st['pcg_iters'][:max(d,1)]))}/{max(st['pcg_iters'][:max(d,1)])}"
                  + ("" if ok else "  FAILED: " + o.last_error()), flush=True)
print(f"{2 * len(cases)} cases, {bad} failed")
sys.exit(1 if bad else 0)
