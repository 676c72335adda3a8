#!/usr/bin/env python3
"""Microseconds per PCG iteration (the solve phase of every Gauss-Newton iteration divided by its PCG count) and ms of the
linearisation + hierarchy refresh, unprofiled, as optimize(20) runs them (`python scripts/solve_us.py C2 C4`)."""
import sys, os
sys.path.insert(0, os.getcwd())
from sparse_gslam_amd import capi, synth
for name in sys.argv[1:]:
    g = synth.config(name)
    with capi.Optimizer(0) as o:
        o.set_graph(*g.arrays())
        o.optimize(20)
        o.set_poses(g.poses)
        done, st = o.optimize(20)
    its = st["pcg_iters"][:done]
    sol = st["seconds_solve"][:done]
    lin = st["seconds_linearize"][:done]
    print(name, "us per PCG iteration:", [round(1e6 * s / i, 1) for s, i in zip(sol, its)][2:12], "linearize+refresh ms:", [round(1e3 * x, 3) for x in lin][2:8])
