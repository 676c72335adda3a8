#!/usr/bin/env python3
"""Where an optimize(20) of a small chain-like graph spends its time: per-kernel profile + wall clock."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = synth.manhattan(V, V - 1 + NC, seed=1, info_mode="full", init="odom", phi=10.0)
with capi.Optimizer(0, verbose=0) as o:
    for rep in range(3):
        t = time.perf_counter(); o.set_graph(*g.arrays()); t1 = time.perf_counter()
        done, st = o.optimize(20); t2 = time.perf_counter()
        P = o.get_poses(); t3 = time.perf_counter()
        print(f"rep {rep}: set_graph {1e3*(t1-t):.2f} ms, optimize(20) {1e3*(t2-t1):.2f} ms (device {1e3*sum(st['seconds']):.2f}), get_poses {1e3*(t3-t2):.2f} ms; pcg {st['pcg_iters'][:6]}")
with capi.Optimizer(0, profile=1) as o:
    o.set_graph(*g.arrays())
    o.profile_reset()
    o.optimize(20)
    prof = o.kernel_profile()
tot = sum(v["ms"] for v in prof.values())
print(f"profile mode: {tot:.2f} ms of kernels in {sum(v['launches'] for v in prof.values())} launches")
for n, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print(f"  {n:45s} {v['launches']:6d} launches {v['ms']:8.3f} ms  {1e3*v['ms']/v['launches']:7.2f} us")
