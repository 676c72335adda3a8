#!/usr/bin/env python3
"""optimize(20) on a spread of synthetic graphs: PCG iteration counts, time per GN iteration and (for
the sizes the CPU oracle solves in seconds) agreement with the direct-solver oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import c_oracle  # noqa: E402
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = [
    ("V=2k E=2.2k seed 1", dict(V=2000, E=2200, seed=1)),
    ("V=2k E=8k seed 2 full", dict(V=2000, E=8000, seed=2, info_mode="full")),
    ("V=10k E=40k seed 3", dict(V=10000, E=40000, seed=3)),
    ("V=10k E=40k seed 4 full phi=10", dict(V=10000, E=40000, seed=4, info_mode="full", phi=10.0)),
    ("V=10k E=100k seed 5", dict(V=10000, E=100000, seed=5)),
    ("V=30k E=33k seed 6", dict(V=30000, E=33000, seed=6)),
    ("V=100k E=1M seed 9 full", dict(V=100000, E=1000000, seed=9, info_mode="full")),
]
for name, kw in cases:
    g = synth.manhattan(**kw)
    with capi.Optimizer(0) as o:
        o.set_graph(*g.arrays())
        done, st = o.optimize(20)
    line = (f"{name:34s} done={done} pcg min/mean/max={min(st['pcg_iters'])}/{np.mean(st['pcg_iters']):.1f}/{max(st['pcg_iters'])} "
            f"gn_ms={1e3 * np.median(st['seconds']):.2f} chi2 {st['chi2'][0]:.4g}->{st['chi2'][-1]:.6g}")
    if g.V <= 10000:
        t = time.time()
        _, ost = c_oracle.gauss_newton(*g.arrays(), iters=20)
        rel = max(abs(a - b) / b for a, b in zip(st["chi2"], ost["chi2"]))
        line += f"  max rel chi2 diff vs direct oracle {rel:.1e} (oracle {time.time() - t:.1f}s)"
    print(line, flush=True)
