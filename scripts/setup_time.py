#!/usr/bin/env python3
"""Time sgo_set_graph_se2 (host structure build + upload + multigrid set-up) and optimize(20)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

for name in sys.argv[1].split(","):
    g = synth.config(name)
    with capi.Optimizer(0) as o:
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            o.set_graph(*g.arrays())
            ts.append(time.perf_counter() - t)
        t = time.perf_counter()
        done, st = o.optimize(20)
        to = time.perf_counter() - t
    print(f"{name}: set_graph {1e3 * min(ts):.1f} ms (first {1e3 * ts[0]:.1f}), optimize(20) {1e3 * to:.1f} ms, "
          f"device GN sum {1e3 * sum(st['seconds']):.1f} ms", flush=True)
