#!/usr/bin/env python3
"""Where do the extra PCG iterations of bench.py's incremental_session come from (22-27 per solve against 18-24 after a fresh
set-up)?  The same session twice: (A) as bench.py runs it -- the resident hierarchy aggregated at the base graph's INITIAL poses --,
(B) with one more sgo_set_graph_se2 of the base graph at its OPTIMISED poses before the first update (what the resident hierarchy is
in the reference's steady state: every full set-up happens at converged poses).  python scripts/incremental_staleness_probe.py [V E]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps, chain, iters = 12, 25, 20
base, app, g = synth.append_session(V, E, steps, chain, 4)
odom_meas = g.meas[: g.V - 1]
for variant in ("A: hierarchy from the initial poses", "B: hierarchy from the optimised poses"):
    arrs = [base.ei, base.ej, base.meas, base.info, base.phi]
    with capi.Optimizer(0) as inc, capi.Optimizer(0) as fresh:
        inc.set_graph(*base.arrays())
        inc.optimize(iters)
        P = inc.get_poses()
        if variant.startswith("B"):
            inc.set_graph(P, base.fixed, *arrs)
            inc.optimize(iters)
            P = inc.get_poses()
        E_res = base.E
        t_opt, t_fopt, its, fits = [], [], [], []
        for k, a in enumerate(app):
            arrs = [np.concatenate([x, a[n]]) for x, n in zip(arrs, ("ei", "ej", "meas", "info", "phi"))]
            P0 = np.empty((a["V"], 3))
            P0[: P.shape[0]] = P
            synth.chain_init(P0, odom_meas, P.shape[0], a["V"] - 1)
            fixed = np.zeros(a["V"], dtype=bool)
            fixed[0] = True
            inc.update_graph(P0, fixed, *arrs, E_res)
            t = time.perf_counter(); d, st = inc.optimize(iters); t_opt.append(1e3 * (time.perf_counter() - t))
            P = inc.get_poses()
            its.append(float(np.mean(st["pcg_iters"][:iters])))
            fresh.set_graph(P0, fixed, *arrs)
            t = time.perf_counter(); df, sf = fresh.optimize(iters); t_fopt.append(1e3 * (time.perf_counter() - t))
            fits.append(float(np.mean(sf["pcg_iters"][:iters])))
            E_res = arrs[0].size
        print(f"{variant}: optimize(20) median {np.median(t_opt):.1f} ms, PCG per solve {np.mean(its):.1f}; after a fresh set-up {np.median(t_fopt):.1f} ms, {np.mean(fits):.1f}",
              flush=True)
