"""The multifrontal path against the multigrid PCG over graph sizes and closure densities (one GPU).

    python scripts/mfront_sizes.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402


def run(g, env, iters=20, reps=3):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with capi.Optimizer(0, direct_rows=1) as opt:
            t0 = time.perf_counter()
            opt.set_graph(*g.arrays())
            ts = time.perf_counter() - t0
            d = opt.solver_description()
            best = 1e9
            for _ in range(reps):
                opt.set_poses(g.poses)
                t0 = time.perf_counter()
                rc, st = opt.optimize(iters)
                best = min(best, time.perf_counter() - t0)
            return d, ts, best / iters, st
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


for V, E in [(1000, 1300), (2000, 2800), (5489, 7629), (8000, 9000), (8000, 12000), (16000, 21000), (16000, 28000), (30000, 40000),
             (30000, 55000)]:
    g = synth.manhattan(V, E, seed=7, info_mode="full", phi=0.75)
    da, tsa, ta, sa = run(g, {"SGO_MFRONT": "1"})
    db, tsb, tb, sb = run(g, {"SGO_MFRONT": "0"})
    rel = abs(sa["chi2"][-1] - sb["chi2"][-1]) / sb["chi2"][-1]
    print(f"V={V:6d} E={E:6d}  {da.split(':')[0]:22s} set-up {1e3 * tsa:6.1f} ms  {1e3 * ta:7.3f} ms/GN it   |  {db.split(':')[0]:8s} set-up {1e3 * tsb:6.1f} ms "
          f"{1e3 * tb:7.3f} ms/GN it ({np.mean(sb['pcg_iters']):.0f} PCG its)   chi2 rel diff {rel:.1e}")
    if da.startswith("multifrontal"):
        print("      ", da.split(";")[0][:230])
    else:
        print("      ", da.split("multifrontal path not used:")[-1][:200])
