#!/usr/bin/env python3
"""Growth sessions (graph set up at its initial poses, optimised, then closures appended through sgo_update_graph_se2): PCG counts
and optimize(20) times per update with the round-5 rules (lagged refresh + re-aggregation when the blocks have moved far since the
hierarchy was made) and without (SGO_AMG_LAG=0), and after a fresh set-up of the same arrays.  python scripts/agg_rule_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

os.environ["SGO_MFRONT"] = "0"
for V, E in ((10000, 40000), (20000, 200000), (30000, 300000), (50000, 250000), (100000, 1000000)):
    base, app, g = synth.append_session(V, E, 4, 25, 4)
    odom_meas = g.meas[: g.V - 1]
    for lag in ("1", "0"):
        os.environ["SGO_AMG_LAG"] = lag
        arrs = [base.ei, base.ej, base.meas, base.info, base.phi]
        rows = []
        with capi.Optimizer(0, direct_rows=0) as inc, capi.Optimizer(0, direct_rows=0) as fresh:
            inc.set_graph(*base.arrays())
            d, st = inc.optimize(20); P = inc.get_poses(); E_res = base.E
            for k, a in enumerate(app):
                arrs = [np.concatenate([x, a[n]]) for x, n in zip(arrs, ("ei", "ej", "meas", "info", "phi"))]
                P0 = np.empty((a["V"], 3)); P0[: P.shape[0]] = P
                synth.chain_init(P0, odom_meas, P.shape[0], a["V"] - 1)
                fixed = np.zeros(a["V"], dtype=bool); fixed[0] = True
                inc.update_graph(P0, fixed, *arrs, E_res)
                t = time.perf_counter(); d, st = inc.optimize(20); ti = 1e3 * (time.perf_counter() - t)
                P = inc.get_poses(); E_res = arrs[0].size
                note = "re-aggregated" if "re-aggregated" in inc.solver_description() else ""
                fresh.set_graph(P0, fixed, *arrs)
                t = time.perf_counter(); df, sf = fresh.optimize(20); tf = 1e3 * (time.perf_counter() - t)
                rows.append(f"{ti:.1f} ms / {np.mean(st['pcg_iters'][:20]):.1f} its {note} (fresh {tf:.1f} / {np.mean(sf['pcg_iters'][:20]):.1f})")
        print(f"V={V} E={E} SGO_AMG_LAG={lag}: " + "; ".join(rows), flush=True)
