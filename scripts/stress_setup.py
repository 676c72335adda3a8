#!/usr/bin/env python3
"""Soak test of the set-up variants on graphs large enough for the helper thread (>= 20 000 free poses): random
size / density / long-range share / information shape / extra fixed vertices; every graph through the default set-up
(helper thread, device-made product lists) and through the serial set-up with host-made lists: same hierarchy
description, PCG counts within one, chi2 histories to 1e-7 (1e-4 on long chains, see below)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

# (round 6: the two set-ups differ in the 10th digit -- strengths from the edge list against block norms --, and a solve that keeps
# or refreshes its coarse operators by a threshold rule can fall on either side of it: the counts are compared with the lagged refresh
# off, which is what "within one" was written for)
os.environ["SGO_AMG_LAG"] = "0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for case in range(n_cases):
    V = int(rng.choice([21000, 30000, 45000, 60000]))
    dens = float(rng.choice([1.05, 2.0, 5.0, 10.0]))
    kw = dict(V=V, E=int(dens * V), seed=int(rng.integers(1, 10**6)), p_random=float(rng.choice([0.0, 0.0, 0.02, 0.2])),
              info_mode=str(rng.choice(["diag", "full"])), phi=float(rng.choice([1.0, 10.0])))
    g = synth.manhattan(**kw)
    if rng.random() < 0.5:
        g.fixed[rng.integers(1, V, 4)] = True
    res = []
    for pipe, lists in (("1", "device"), ("0", "host")):
        os.environ["SGO_SETUP_PIPELINE"] = pipe
        os.environ["SGO_AMG_LISTS"] = lists
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            desc = o.solver_description()
            done, st = o.optimize(5)
        res.append((desc, done, st["chi2"], st["pcg_iters"]))
    ok = res[0][0] == res[1][0] and res[0][1] == res[1][1] == 5
    ok = ok and max(abs(a - b) for a, b in zip(res[0][3], res[1][3])) <= 1
    rel = max(abs(a - b) / b for a, b in zip(res[0][2], res[1][2]))
    # long chains (fewer than 1.5 edges per pose): kappa(H) grows with the square of the chain length and two solves
    # that differ by rounding already differ by 1e-6 and more in the transient chi2 of undamped Gauss-Newton
    # (DESIGN.md section 5a); everything else must agree to 1e-7
    ok = ok and rel <= (1e-4 if kw["E"] < 1.5 * V else 1e-7)
    print(f"{case:2d} V={V} E={kw['E']} {kw['info_mode']} p_rand={kw['p_random']:.2f} rel={rel:.1e} pcg={res[0][3]} / {res[1][3]}"
          + ("" if ok else "   <-- MISMATCH\n    " + res[0][0][:200] + "\n    " + res[1][0][:200]), flush=True)
    bad += 0 if ok else 1
print(f"{n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
