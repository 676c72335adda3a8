#!/usr/bin/env python3
"""Sweep of the multigrid's scalar parameters (SGO_AMG_OMEGA smoother damping, SGO_AMG_OMEGA_P prolongator smoothing,
SGO_AMG_THETA strength threshold) on the bench configs: total PCG iterations and time of optimize(20)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["C4", "C2", "C3s"]
graphs = {n: synth.config(n) for n in names}
grid = [("SGO_AMG_OMEGA", v) for v in ("0.7", "0.8", "0.9", "1.0")] + [("SGO_AMG_OMEGA_P", v) for v in ("0.5", "0.66", "0.8", "1.0")] + \
       [("SGO_AMG_THETA", v) for v in ("0.01", "0.02", "0.04", "0.08")]
for key, val in grid:
    os.environ[key] = val
    row = []
    for n in names:
        g = graphs[n]
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            o.optimize(20)
            o.set_poses(g.poses)
            done, st = o.optimize(20)
        row.append(f"{n}: {sum(st['pcg_iters'][:done]):4d} its {1e3 * sum(st['seconds'][:done]):6.1f} ms (done {done})")
    print(f"{key}={val:5s} " + " | ".join(row), flush=True)
    del os.environ[key]
