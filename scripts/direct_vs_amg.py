#!/usr/bin/env python3
"""optimize(20) of chain-like graphs through the single-launch direct path and through the multigrid PCG path."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = [(int(a), int(b)) for a, b in (c.split("/") for c in sys.argv[1:])] or [(1000, 30), (4000, 40), (8000, 45), (16000, 48)]
for V, NC in cases:
    g = synth.manhattan(V, V - 1 + NC, seed=1, info_mode="full", init="odom", phi=10.0)
    out = []
    for rows in (1 << 20, 0):
        with capi.Optimizer(0, direct_rows=rows) as o:
            ts = []
            for rep in range(3):
                t = time.perf_counter(); o.set_graph(*g.arrays()); t1 = time.perf_counter()
                done, st = o.optimize(20); t2 = time.perf_counter()
                ts.append((t1 - t, t2 - t1))
            out.append((o.solver_description().split(":")[0], min(a for a, _ in ts), min(b for _, b in ts), done, st["chi2"][-1], o.get_poses()))
    (d1, s1, o1, k1, c1, P1), (d2, s2, o2, k2, c2, P2) = out
    print(f"V={V} closures={NC}: {d1}: set_graph {1e3*s1:.2f} ms optimize(20) {1e3*o1:.2f} ms (done {k1}) | {d2}: set_graph {1e3*s2:.2f} ms "
          f"optimize(20) {1e3*o2:.2f} ms (done {k2}) | final chi2 rel diff {abs(c1-c2)/c2:.1e} poses {np.abs(P1-P2).max():.1e}", flush=True)
