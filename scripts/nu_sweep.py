#!/usr/bin/env python3
"""Coarse-level sweeps per side (SGO_AMG_NU = 1 / 2) over graph sizes: ms per GN iteration and PCG iterations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from sparse_gslam_amd import capi, synth  # noqa: E402

for V, E in ((10000, 40000), (20000, 200000), (30000, 300000), (50000, 250000), (50000, 500000), (70000, 700000)):
    g = synth.manhattan(V, E, seed=7)
    out = []
    for nu in ("1", "2"):
        os.environ["SGO_AMG_NU"] = nu
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            o.optimize(20)
            o.set_poses(g.poses)
            done, st = o.optimize(20)
        out.append((float(np.median(st["seconds"][2:done])) * 1e3, sum(st["pcg_iters"][:done]) / done, 1e3 * sum(st["seconds"][:done])))
    print(f"V={V} E={E}: nu=1 {out[0][0]:.3f} ms/GN ({out[0][1]:.1f} its, optimize {out[0][2]:.1f} ms)   nu=2 {out[1][0]:.3f} ms/GN ({out[1][1]:.1f} its, optimize {out[1][2]:.1f} ms)", flush=True)
