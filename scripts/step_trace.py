#!/usr/bin/env python3
"""Per-GN-iteration times and PCG counts of repeated optimize(20) calls from the same start: `python scripts/step_trace.py C4 [calls]`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = synth.config(name)
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    print(o.solver_description(), flush=True)
    for k in range(calls):
        o.set_poses(g.poses)
        t = time.perf_counter()
        done, st = o.optimize(20)
        dt = time.perf_counter() - t
        ms = [round(1e3 * x, 2) for x in st["seconds"][:done]]
        print(f"call {k}: {1e3 * dt:.1f} ms, done {done}, pcg {st['pcg_iters']}, sum {sum(st['pcg_iters'])}\n   ms per GN iteration {ms}", flush=True)
