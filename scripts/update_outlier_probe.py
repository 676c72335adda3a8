#!/usr/bin/env python3
"""sgo_update_graph_se2 takes 0.3-0.5 ms, and now and then 16-30 ms.  The same session three ways: as bench.py runs it; with a
trivial device call (sgo_chi2) right before each update; and with the host-side preparation of the next arrays (numpy concatenation,
tens of ms during which the device idles) done BEFORE the previous optimize() instead of after it.  If the outliers move to whichever
device call comes first after the idle period, they are the device's wake-up, not the update's.  python scripts/update_outlier_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

base, app, g = synth.append_session(100000, 1000000, 12, 25, 4)
odom_meas = g.meas[: g.V - 1]
for mode in ("plain", "chi2 first"):
    arrs = [base.ei, base.ej, base.meas, base.info, base.phi]
    t_up, t_pre = [], []
    with capi.Optimizer(0) as inc:
        inc.set_graph(*base.arrays())
        inc.optimize(20)
        P = inc.get_poses()
        E_res = base.E
        for a in app:
            arrs = [np.concatenate([x, a[n]]) for x, n in zip(arrs, ("ei", "ej", "meas", "info", "phi"))]
            P0 = np.empty((a["V"], 3))
            P0[: P.shape[0]] = P
            synth.chain_init(P0, odom_meas, P.shape[0], a["V"] - 1)
            fixed = np.zeros(a["V"], dtype=bool)
            fixed[0] = True
            if mode == "chi2 first":
                t = time.perf_counter(); inc.chi2(); t_pre.append(round(1e3 * (time.perf_counter() - t), 2))
            t = time.perf_counter(); inc.update_graph(P0, fixed, *arrs, E_res); t_up.append(round(1e3 * (time.perf_counter() - t), 2))
            inc.optimize(20)
            P = inc.get_poses()
            E_res = arrs[0].size
    print(f"{mode:12s} update ms {t_up}" + (f"\n             the sgo_chi2 before it {t_pre}" if t_pre else ""), flush=True)
