#!/usr/bin/env python3
"""Soak of the incremental set-up (sgo_update_graph_se2): random growth sessions -- chain lengths 1..80, 0..3 closures per
step that end in resident poses, in poses of earlier updates (hubs), in the fixed vertex, duplicates -- every optimize() after
an update against a FRESH sgo_set_graph_se2 of the same arrays from the same initial poses (bound: BASELINE.json's 1e-6 on
every iterate's chi2 and robust chi2, 1e-5 m on poses).  Prints one line per session and a summary.
Usage: python scripts/overlay_stress.py [sessions] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402


def session(k, seed):
    rng = np.random.default_rng(seed * 7919 + k)
    V0 = int(rng.integers(2500, 14000))
    epv = float(rng.uniform(1.4, 6.0))
    steps = int(rng.integers(3, 9))
    chains = [int(rng.integers(1, 81)) for _ in range(steps)]
    Vt = V0 + sum(chains)
    info_mode = str(rng.choice(["diag", "full"]))
    phi = float(rng.choice([1.0, 10.0]))
    g = synth.manhattan(Vt, int(Vt * epv), int(rng.integers(1, 1 << 30)), 0.0, info_mode=info_mode, phi=phi, init="incremental", tail=0)
    keep = (g.ei < V0) & (g.ej < V0)
    arrs = [g.ei[keep], g.ej[keep], g.meas[keep], g.info[keep], g.phi[keep]]
    odom_meas = g.meas[: Vt - 1]
    sig = np.array([synth.SIGMA_XY, synth.SIGMA_XY, synth.SIGMA_TH])

    def closure(i, j):
        z = synth._rel(g.truth[[i]], g.truth[[j]])[0] + rng.standard_normal(3) * sig
        z[2] = synth._wrap(z[2])
        return i, j, z, np.array([1 / sig[0]**2, 0, 0, 1 / sig[1]**2, 0, 1 / sig[2]**2]), phi

    worst, n_inc, n_full, hubs, its = 0.0, 0, 0, 0, []
    iters = 6
    with capi.Optimizer(0, direct_rows=0) as inc, capi.Optimizer(0, direct_rows=0) as fresh:
        fixed = np.zeros(V0, dtype=bool)
        fixed[0] = True
        inc.set_graph(g.poses[:V0], fixed, *arrs)
        inc.optimize(iters)
        P = inc.get_poses()
        a = V0
        for ch in chains:
            b = a + ch
            new = [(i, i + 1, odom_meas[i], g.info[i], -1.0) for i in range(a - 1, b - 1)]
            for _ in range(int(rng.integers(0, 4))):
                src = int(rng.integers(a, b))
                r = rng.random()
                if r < 0.55:
                    tgt = int(rng.integers(1, V0))                      # a resident pose
                elif r < 0.85 and a > V0:
                    tgt = int(rng.integers(V0, a))                      # a pose of an earlier update
                elif r < 0.93:
                    tgt = 0                                             # the fixed vertex
                else:
                    tgt = int(rng.integers(a, b))                       # inside this chain
                if tgt == src:
                    continue
                c = closure(min(src, tgt), max(src, tgt)) if rng.random() < 0.5 else closure(max(src, tgt), min(src, tgt))
                new.append(c)
                if rng.random() < 0.15:
                    new.append(c)                                       # a duplicate
            E_res = arrs[0].size
            arrs = [np.concatenate([arrs[0], np.array([e[0] for e in new], np.int32)]),
                    np.concatenate([arrs[1], np.array([e[1] for e in new], np.int32)]),
                    np.concatenate([arrs[2], np.array([e[2] for e in new])]),
                    np.concatenate([arrs[3], np.array([e[3] for e in new])]),
                    np.concatenate([arrs[4], np.array([e[4] for e in new])])]
            P0 = np.empty((b, 3))
            P0[:a] = P
            synth.chain_init(P0, odom_meas, a, b - 1)
            fixed = np.zeros(b, dtype=bool)
            fixed[0] = True
            inc.update_graph(P0, fixed, *arrs, E_res)
            d = inc.solver_description()
            if "incremental overlay" in d:
                n_inc += 1
                hubs = max(hubs, int(d.split("appended rows (")[1].split(" hubs")[0]))
            else:
                n_full += 1
            d1, s1 = inc.optimize(iters)
            P = inc.get_poses()
            fresh.set_graph(P0, fixed, *arrs)
            d2, s2 = fresh.optimize(iters)
            Pf = fresh.get_poses()
            if d1 != d2:
                worst = float("inf")
            elif d1 == iters:
                worst = max(worst, max(abs(x - y) / y for x, y in zip(s1["chi2"], s2["chi2"])),
                            max(abs(x - y) / y for x, y in zip(s1["robust_chi2"], s2["robust_chi2"])))
                if np.abs(P - Pf).max() > 1e-5:
                    worst = max(worst, float(np.abs(P - Pf).max()))
                its.append(np.mean(s1["pcg_iters"]) - np.mean(s2["pcg_iters"]))
            a = b
    return dict(V0=V0, E0=int(keep.sum()), steps=steps, chains=chains, info=info_mode, phi=phi, worst=worst, incremental=n_inc, full=n_full,
                hubs=hubs, extra_its=float(np.mean(its)) if its else 0.0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad, tot_inc, tot_full = 0, 0, 0
    for k in range(n):
        t0 = time.time()
        r = session(k, seed)
        ok = r["worst"] <= 1e-6
        bad += not ok
        tot_inc += r["incremental"]
        tot_full += r["full"]
        print(f"session {k}: V0={r['V0']} E0={r['E0']} {r['info']} phi={r['phi']} chains={r['chains']}: {'ok' if ok else 'MISMATCH'}; worst rel diff "
              f"{r['worst']:.1e}; updates incremental / full {r['incremental']} / {r['full']}; up to {r['hubs']} hubs; PCG iterations per solve vs fresh "
              f"{r['extra_its']:+.1f} ({time.time() - t0:.1f} s)", flush=True)
    print(f"{n} sessions, {bad} bad; {tot_inc} incremental updates, {tot_full} full set-ups", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
