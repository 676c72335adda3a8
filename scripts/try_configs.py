#!/usr/bin/env python3
"""Run optimize(iters) on a few synthetic graphs and print solver behaviour (exploration helper)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = {
    "chain100k": dict(V=100_000, E=110_000, seed=7, p_random=0.0),
    "C4r": dict(synth.CONFIGS["C4r"]),
    "C5": dict(synth.CONFIGS["C5"]),
    "C5local": dict(V=1_000_000, E=10_000_000, seed=5, p_random=0.0),
}
names = sys.argv[1].split(",")
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for name in names:
    t = time.time()
    g = synth.manhattan(**cases[name])
    tg = time.time() - t
    with capi.Optimizer(0, pcg_maxit=3000, verbose=1) as o:
        t = time.time()
        o.set_graph(*g.arrays())
        ts = time.time() - t
        done, st = o.optimize(iters)
    print(f"{name}: gen {tg:.1f}s setup {ts:.2f}s done={done} pcg={st['pcg_iters']} "
          f"gn_ms={[round(1e3 * s, 1) for s in st['seconds']]} chi2={st['chi2'][0]:.6g}->{st['chi2'][-1]:.6g}", flush=True)
