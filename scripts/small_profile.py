#!/usr/bin/env python3
"""Where the time goes on a small (reference-scale) graph: set_graph / optimize(20) wall time,
per-phase seconds from sgo_stats and the per-kernel profile.   python scripts/small_profile.py V E"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
E = int(sys.argv[2]) if len(sys.argv) > 2 else V + 30
g = synth.manhattan(V, E, seed=1, info_mode="full", init="odom", phi=10.0)
for prof in (0, 1):
    with capi.Optimizer(0, profile=prof) as opt:
        for rep in range(3):
            t0 = time.perf_counter()
            opt.set_graph(g.poses, g.fixed, g.ei, g.ej, g.meas, g.info, g.phi)
            t1 = time.perf_counter()
            done, st = opt.optimize(20)
            t2 = time.perf_counter()
            P = opt.get_poses()
            t3 = time.perf_counter()
        print(f"profile={prof} V={V} E={E}: set_graph {1e3*(t1-t0):.2f} ms  optimize(20) {1e3*(t2-t1):.2f} ms  get_poses {1e3*(t3-t2):.2f} ms")
        print("  pcg_iters", list(st["pcg_iters"][:done]))
        print(f"  seconds: total {1e3*st['seconds_total']:.2f} ms setup {1e3*st['seconds_setup']:.2f} ms "
              f"lin {1e3*np.sum(st['seconds_linearize'][:done]):.2f} ms solve {1e3*np.sum(st['seconds_solve'][:done]):.2f} ms")
        if prof:
            rows = opt.kernel_profile()
            tot = sum(r["ms"] for r in rows.values())
            print(f"  kernel time total {tot:.2f} ms over {sum(r['launches'] for r in rows.values())} launches")
            for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:12]:
                print(f"    {name:<28s} launches {r['launches']:6d}  total {r['ms']:8.3f} ms  avg {1e3*r['ms']/r['launches']:7.2f} us")
