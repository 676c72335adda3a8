#!/usr/bin/env python3
"""The reference's flow on a config: sgo_set_graph_se2 + optimize(20), four cycles on one context -- ms of each part and the
first Gauss-Newton iterations' times (`python scripts/cycle_time.py C4 C2`)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from sparse_gslam_amd import capi, synth
for name in sys.argv[1:]:
    g = synth.config(name)
    with capi.Optimizer(0) as o:
        for k in range(4):
            t = time.perf_counter(); o.set_graph(*g.arrays()); t1 = time.perf_counter()
            done, st = o.optimize(20); t2 = time.perf_counter()
            print(f"{name} cycle {k}: set_graph {1e3*(t1-t):.1f} ms, optimize(20) {1e3*(t2-t1):.1f} ms, first GN iterations {[round(1e3*x,2) for x in st['seconds'][:3]]} chi2 {st['chi2'][done]:.9g}", flush=True)
