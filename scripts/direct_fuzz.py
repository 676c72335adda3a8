#!/usr/bin/env python3
"""Randomised check of the single-launch direct path against the CPU oracle: chain graphs of random length with random
closures (some doubled, some between closure endpoints), random fixed poses, random gaps in the chain (components
anchored by a fixed pose each)."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import c_oracle  # noqa: E402
from sparse_gslam_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = direct = 0
with capi.Optimizer(0) as o:
    for case in range(N):
        V = int(rng.integers(2, 3000)) if case % 4 else int(rng.integers(2, 70))
        nc = int(rng.integers(0, min(58, max(1, V // 3)) + 1))
        g = synth.manhattan(V, V - 1 + nc, seed=int(rng.integers(1 << 30)), info_mode="full" if case % 2 else "diag",
                            init="odom" if case % 3 else "incremental", phi=10.0)
        ei, ej, meas, info, phi = [a.copy() for a in (g.ei, g.ej, g.meas, g.info, g.phi)]
        fixed = g.fixed.copy()
        if nc and case % 5 == 0:          # double some closures
            d = rng.choice(np.arange(V - 1, g.E), size=min(3, nc), replace=False)
            ei, ej = np.concatenate([ei, ei[d]]), np.concatenate([ej, ej[d]])
            meas, info, phi = np.concatenate([meas, meas[d] + 0.01]), np.concatenate([info, info[d]]), np.concatenate([phi, phi[d]])
        if V > 10 and case % 3 == 0:      # a few more fixed poses
            fixed[rng.choice(V, size=int(rng.integers(1, 4)), replace=False)] = True
        if V > 20 and case % 7 == 0:      # cut the chain once; anchor the second part
            cut = int(rng.integers(5, V - 5))
            keep = ~(((ei == cut) & (ej == cut + 1)) | ((ei == cut + 1) & (ej == cut)))
            ei, ej, meas, info, phi = ei[keep], ej[keep], meas[keep], info[keep], phi[keep]
            fixed[cut + 1] = True
        args = [g.poses, fixed, ei, ej, meas, info, phi]
        o.set_graph(*args)
        is_direct = o.solver_description().startswith("direct_ldlt")
        direct += is_direct
        done, st = o.optimize(10)
        P = o.get_poses()
        oP, ost = c_oracle.gauss_newton(*args, iters=10)
        ok = done == ost["iters_done"]
        rel = 0.0
        if ok and done > 0:
            # relative to the iterate's own chi2, with a floor for chains that converge to chi2 = 0 (no closures)
            rel = max(abs(st["chi2"][k] - ost["chi2"][k]) / max(ost["chi2"][k], 1e-9 * ost["chi2"][0], 1e-9) for k in range(done + 1))
            ok = rel < 1e-6 and np.abs(P - oP).max() < 1e-5          # BASELINE.json's bound
        if not ok:
            bad += 1
            print(f"case {case}: V={V} nc={nc} direct={is_direct} done {done} vs {ost['iters_done']} rel {rel:.2e} pose {np.abs(P - oP).max():.2e}", flush=True)
print(f"{N} cases, {direct} on the direct path, {bad} bad")
