#!/usr/bin/env python3
"""Round 6: sgo_set_graph_se2 and optimize(20) with the multigrid set-up's patterns made on the host (SGO_AMG_SETUP=host), on the
device for the rebuilds inside a call only (=rebuilds) and on the device for every set-up (=device; the helper thread of the set-up
pipeline then makes the aggregation alone).  Median of 7 set_graph calls on one context (steady state: arenas grown).
Usage: python scripts/setup_modes.py C2,C4,C4:odom,C4r,C5"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

for spec in (sys.argv[1] if len(sys.argv) > 1 else "C2,C4,C4:odom").split(","):
    name, _, init = spec.partition(":")
    g = synth.config(name, **({"init": init} if init else {}))
    for mode in ("host", "rebuilds", "device"):
        os.environ["SGO_AMG_SETUP"] = mode
        with capi.Optimizer(0) as o:
            ts = []
            for _ in range(7):
                t = time.perf_counter()
                o.set_graph(*g.arrays())
                ts.append(time.perf_counter() - t)
            t = time.perf_counter()
            done, st = o.optimize(20)
            to = time.perf_counter() - t
            o.set_graph(*g.arrays())
            t = time.perf_counter()
            done, st = o.optimize(20)
            to2 = time.perf_counter() - t
        print(f"{spec:9s} {mode:9s}: set_graph median {1e3 * float(np.median(ts)):7.1f} ms (min {1e3 * min(ts):.1f}, first {1e3 * ts[0]:.1f}); optimize(20) "
              f"{1e3 * to:7.1f} / {1e3 * to2:7.1f} ms, pcg {int(sum(st['pcg_iters']))}, final chi2 {st['chi2'][-1]!r}", flush=True)
