#!/usr/bin/env python3
"""Lagged refresh of the coarse operators (sgo_solve.cpp) on and off, in one process, over graph shapes: optimize(20) from the
incremental start, best of three passes each.  python scripts/lag_probe.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ["SGO_MFRONT"] = "0"
SHAPES = [
    ("C4", synth.config("C4")),
    ("C2", synth.config("C2")),
    ("C3s", synth.config("C3s")),
    ("C4 full information phi=10", synth.config("C4", info_mode="full", phi=10.0)),
    ("20k / 60k", synth.manhattan(20000, 60000, seed=21)),
    ("50k / 250k", synth.manhattan(50000, 250000, seed=22)),
    ("50k / 250k full phi=10", synth.manhattan(50000, 250000, seed=22, info_mode="full", phi=10.0)),
    ("30k / 300k", synth.manhattan(30000, 300000, seed=23)),
    ("100k / 400k", synth.manhattan(100000, 400000, seed=24)),
    ("50k / 250k 1 % random", synth.manhattan(50000, 250000, seed=25, p_random=0.01)),
]
for name, g in SHAPES:
    row = []
    for lag in ("0", "1"):
        os.environ["SGO_AMG_LAG"] = lag
        best, its, kept = 1e9, None, ""
        with capi.Optimizer(0, direct_rows=0) as o:
            for _ in range(3):
                o.set_graph(*g.arrays())
                t = time.perf_counter()
                d, st = o.optimize(iters)
                dt = time.perf_counter() - t
                assert d == iters, o.last_error()
                if dt < best:
                    best, its = dt, st["pcg_iters"][:iters]
                desc = o.solver_description()
                kept = desc.split("last sgo_optimize_gn: ")[1].split(" solves")[0] if "last sgo_optimize_gn: " in desc else "0"
        row.append((best, sum(its), kept, its))
    (t0, s0, _, i0), (t1, s1, k1, i1) = row
    print(f"{name:28s} V={g.V:7d} E={g.E:8d}  refresh always {1e3 * t0:7.1f} ms ({s0} PCG iterations)   lagged {1e3 * t1:7.1f} ms ({s1}; kept {k1})"
          f"   {100.0 * (t0 / t1 - 1.0):+5.1f} %", flush=True)
    print("    always", i0)
    print("    lagged", i1, flush=True)
