#!/usr/bin/env python3
"""Graphs whose multigrid hierarchy falls back to the tentative prolongator (random long-range closures): how many
flexible-CG steps the K-cycle takes per level (SGO_AMG_FCG2_DEPTH: levels deeper than this take one step).
Prints optimize(20) time, median GN iteration and the PCG counts per setting."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = [
    ("2k/8k p=0.05", lambda: synth.manhattan(2000, 8000, seed=31, p_random=0.05, info_mode="full")),
    ("8k/32k p=0.05", lambda: synth.manhattan(8000, 32000, seed=11, p_random=0.05, info_mode="full")),
    ("10k/40k p=0.2", lambda: synth.manhattan(10000, 40000, seed=32, p_random=0.2)),
    ("30k/120k p=0.02", lambda: synth.manhattan(30000, 120000, seed=33, p_random=0.02)),
    ("30k/300k p=0.1", lambda: synth.manhattan(30000, 300000, seed=34, p_random=0.1, info_mode="full")),
    ("C4r", lambda: synth.config("C4r")),
]
if len(sys.argv) > 1 and sys.argv[1] == "C5":
    cases = [("C5", lambda: synth.config("C5"))]
settings = os.environ.get("SGO_SWEEP", "|SGO_AMG_FCG2_DEPTH=2|SGO_AMG_FCG2_DEPTH=1|SGO_AMG_FCG2_DEPTH=0").split("|")
for name, make in cases:
    g = make()
    for setting in settings:
        kv = dict(s.split("=", 1) for s in setting.split())
        os.environ.update(kv)
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            t = time.perf_counter()
            done, st = o.optimize(20)
            to = time.perf_counter() - t
            nl = o.solver_description().count(" n=")
        for k in kv:
            del os.environ[k]
        gn = [1e3 * x for x in st["seconds"][:done]]
        print(f"{name:16s} [{setting or 'default':22s}] levels {nl} done={done} optimize(20) {1e3 * to:8.1f} ms  GN median {statistics.median(gn):7.2f} ms  "
              f"pcg sum {sum(st['pcg_iters'][:done]):5d} max {max(st['pcg_iters'][:done]):4d}  chi2 {st['chi2'][done]:.10e}", flush=True)
