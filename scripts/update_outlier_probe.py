#!/usr/bin/env python3
"""sgo_update_graph_se2 takes 0.3-0.5 ms, and now and then 15-35 ms.  The same growth session three ways:
  plain        as bench.py ran it up to round 5: the arrays of the grown graph are fresh numpy arrays every closure (np.concatenate:
               ~100 MB allocated and as much freed -- mmap / munmap -- per closure)
  chi2 first   the same with a trivial device call (sgo_chi2) right before each update
  in place     the arrays allocated once at their final size and filled in place (and the host away for 50 ms all the same)
Result: the outliers move to whichever device call comes first ("chi2 first"), and they are gone when the host process does not map
and unmap large regions between two calls ("in place"): the driver's work behind munmap / mmap of a process with device queues lands
on the next submission.  Not the update, not the device idling.  python scripts/update_outlier_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

base, app, g = synth.append_session(100000, 1000000, 12, 25, 4)
odom_meas = g.meas[: g.V - 1]
NAMES = ("ei", "ej", "meas", "info", "phi")
Vmax, Emax = app[-1]["V"], base.E + sum(len(a["ei"]) for a in app)
for mode in ("plain", "chi2 first", "in place"):
    t_up, t_pre = [], []
    with capi.Optimizer(0) as inc:
        inc.set_graph(*base.arrays())
        inc.optimize(20)
        P = inc.get_poses()
        E_res = base.E
        arrs = [base.ei, base.ej, base.meas, base.info, base.phi]
        if mode == "in place":
            bufs = [np.empty((Emax,) + x.shape[1:], dtype=x.dtype) for x in arrs]
            for b, x in zip(bufs, arrs):
                b[: base.E] = x
            Pbuf, fbuf, ne = np.empty((Vmax, 3)), np.zeros(Vmax, dtype=bool), base.E
            fbuf[0] = True
        for a in app:
            if mode == "in place":
                k = len(a["ei"])
                for b, n in zip(bufs, NAMES):
                    b[ne: ne + k] = a[n]
                ne += k
                arrs = [b[:ne] for b in bufs]
                P0, fixed = Pbuf[: a["V"]], fbuf[: a["V"]]
                time.sleep(0.05)
            else:
                arrs = [np.concatenate([x, a[n]]) for x, n in zip(arrs, NAMES)]
                P0 = np.empty((a["V"], 3))
                fixed = np.zeros(a["V"], dtype=bool)
                fixed[0] = True
            P0[: P.shape[0]] = P
            synth.chain_init(P0, odom_meas, P.shape[0], a["V"] - 1)
            if mode == "chi2 first":
                t = time.perf_counter(); inc.chi2(); t_pre.append(round(1e3 * (time.perf_counter() - t), 2))
            t = time.perf_counter(); inc.update_graph(P0, fixed, *arrs, E_res); t_up.append(round(1e3 * (time.perf_counter() - t), 2))
            inc.optimize(20)
            P = inc.get_poses()
            E_res = arrs[0].size
    print(f"{mode:12s} update ms {t_up}" + (f"\n             the sgo_chi2 before it {t_pre}" if t_pre else ""), flush=True)
