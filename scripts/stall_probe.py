#!/usr/bin/env python3
"""Round 6: the systems PCG does not finish (scripts/odom_sizes.py's failures: large graphs with full information matrices from the
dead-reckoned start) with the stagnated solve restarted from its own x (SGO_PCG_STALL_RESTARTS).  Usage: python scripts/stall_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = [(150000, 900000, 10), (200000, 2000000, 7)]
for V, E, seed in cases:
    g = synth.manhattan(V, E, seed=seed, init="odom", info_mode="full", phi=10.0)
    for restarts in ("0",):
        os.environ["SGO_PCG_STALL_RESTARTS"] = restarts
        os.environ["SGO_VERBOSE"] = "1"
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            t = time.perf_counter()
            d, st = o.optimize(20)
            to = time.perf_counter() - t
            print(f"V={V} E={E} restarts={restarts}: done {d}, {1e3 * to:.0f} ms, pcg {st['pcg_iters'][:max(d, 1) + 1]}, relres {[f'{r:.1e}' for r in st['pcg_relres'][:max(d,1) + 1]]}"
                  + ("" if d == 20 else "  FAILED: " + o.last_error()), flush=True)
