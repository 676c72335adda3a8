#!/usr/bin/env python3
"""Soak of the round-5 refresh rules (lagged refresh of the coarse operators, re-aggregation across calls) over random graph shapes:
every graph runs optimize(20) three times in a row on one context (set_graph once: calls 2 and 3 start from the optimised poses, as
the reference's repeated optimize(20) does) with the rules on and with SGO_AMG_LAG=0; per graph the chi2 of every iterate of both
runs must agree to 1e-6, and the times of the two are reported.  python scripts/lag_soak.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
os.environ["SGO_MFRONT"] = "0"
bad, ratios = 0, []
for k in range(cases):
    V = int(rng.choice([2000, 5000, 10000, 20000, 40000, 60000]))
    E = int(V * rng.choice([1.5, 2.0, 3.0, 4.0, 6.0, 10.0]))
    kw = dict(seed=int(rng.integers(1, 10000)), info_mode=str(rng.choice(["diag", "full"])), p_random=float(rng.choice([0.0, 0.0, 0.01, 0.05])))
    if kw["info_mode"] == "full":
        kw["phi"] = float(rng.choice([1.0, 10.0]))
    g = synth.manhattan(V, E, **kw)
    out = {}
    for lag in ("1", "0"):
        os.environ["SGO_AMG_LAG"] = lag
        with capi.Optimizer(0, direct_rows=0) as o:
            o.set_graph(*g.arrays())
            chi, t, kept, ok = [], 0.0, [], True
            for call in range(int(os.environ.get("SOAK_CALLS", "3"))):
                t0 = time.perf_counter(); d, st = o.optimize(20); t += time.perf_counter() - t0
                ok = ok and d == 20
                chi += list(st["chi2"][: d + 1])
                desc = o.solver_description()
                kept.append(desc.split("last sgo_optimize_gn: ")[1].split(" of ")[0] if "last sgo_optimize_gn: " in desc else "0")
                if "re-aggregated" in desc:
                    kept[-1] += "+R"
            out[lag] = (np.array(chi), t, kept, ok, "" if ok else o.last_error()[:120])
    (c1, t1, k1, ok1, e1), (c0, t0, _, ok0, e0) = out["1"], out["0"]
    same = ok1 and ok0 and c1.shape == c0.shape and float(np.max(np.abs(c1 - c0) / np.maximum(c0, 1e-300))) <= 1e-6
    if ok1 != ok0 or (ok1 and not same):
        bad += 1
    if ok1 and ok0:
        ratios.append(t0 / t1)
    print(f"{k:3d} V={V:6d} E={E:7d} {kw['info_mode']:4s} p_rand={kw['p_random']:.2f} {'ok ' if (same or not (ok1 or ok0)) else 'BAD'} lagged {1e3 * t1:7.1f} ms (kept {'/'.join(k1)})"
          f"  always {1e3 * t0:7.1f} ms  x{t0 / t1:.2f}" + ("" if ok1 and ok0 else f"  failed: lagged '{e1}' always '{e0}'"), flush=True)
r = np.array(ratios)
print(f"{cases} cases, {bad} bad; time(always refresh) / time(rules on): median {np.median(r):.3f}, min {r.min():.3f}, max {r.max():.3f}, cases slower than 0.97: {(r < 0.97).sum()}")
