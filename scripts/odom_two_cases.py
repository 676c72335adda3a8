#!/usr/bin/env python3
"""The hardest dead-reckoned-start case of scripts/odom_sizes.py alone (200 k poses / 2 M edges, full information, phi 10), for
environment variants: python scripts/odom_two_cases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth
V, E, seed = 200000, 2000000, 7
g = synth.manhattan(V, E, seed=seed, init="odom", info_mode="full", phi=10.0)
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    print(o.solver_description()[:300])
    t = time.perf_counter(); d, st = o.optimize(20); to = time.perf_counter() - t
    print(V, E, "done", d, "optimize ms", round(1e3 * to, 1), "pcg", st["pcg_iters"][:max(d, 1) + 1], "relres", [f"{r:.1e}" for r in st["pcg_relres"][:max(d,1)+1]], "" if d == 20 else o.last_error()[:200], flush=True)
