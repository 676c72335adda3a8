#!/usr/bin/env python3
"""Golden fixture for BASELINE.md's LITERAL start ("dead-reckoned initial guess"): C2 from init="odom".

Undamped Gauss-Newton with the DCS kernel does not converge from there (robust chi2 rises; the reference's algorithm, whatever
solves its systems) and the iteration is chaotic: two correct solvers part ways after a few iterates.  The fixture therefore
records, next to the C++ oracle's direct-solver iterates, HOW MANY iterates the two independent oracles (oracle/sgo_oracle.cpp:
own sparse LDL^T; oracle/np_oracle.py: numpy + SuperLU) agree on to 1e-6 relative in plain AND robust chi2 -- `agree` -- and the
GPU test (tests/test_gpu_golden.py::test_dead_reckoned_start_multigrid_path_matches_the_oracles) compares that many.

Usage: python scripts/make_golden_odom.py [config=C2] [iters=20] [stage=all|cpp|np|merge]
  C2: about a minute of one core.  C4 (round 6): the two oracles as two processes (`cpp` and `np` write their iterates to
  /tmp/<config>_odom_<stage>.npz, `merge` makes the fixture): 13 min + the numpy oracle's SuperLU factorisations.
Writes tests/golden/<config>_odom_direct.npz
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, np_oracle  # noqa: E402
from sparse_gslam_amd import synth  # noqa: E402
from scripts.make_golden_large import graph_digest  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    stage = sys.argv[3] if len(sys.argv) > 3 else "all"
    g = synth.config(name, init="odom")
    part = {k: f"/tmp/{name}_odom_{k}.npz" for k in ("cpp", "np")}

    def run(kind):
        t = time.time()
        mod = c_oracle if kind == "cpp" else np_oracle
        P, s = mod.gauss_newton(*g.arrays(), iters=iters, solver="direct")
        dt = time.time() - t
        np.savez(part[kind], chi2=np.array(s["chi2"]), robust_chi2=np.array(s["robust_chi2"]), seconds=dt,
                 poses_stride50=P[::50].copy())
        print(f"{kind} oracle: {dt:.1f}s, wrote {part[kind]}", flush=True)

    if stage in ("cpp", "np"):
        run(stage)
        return
    if stage == "all":
        run("cpp")
        run("np")
    zc, zn = np.load(part["cpp"]), np.load(part["np"])
    sc = dict(chi2=list(zc["chi2"]), robust_chi2=list(zc["robust_chi2"]))
    sn = dict(chi2=list(zn["chi2"]), robust_chi2=list(zn["robust_chi2"]))
    tc, tn = float(zc["seconds"]), float(zn["seconds"])
    rel = [max(abs(a - b) / abs(b), abs(c - d) / abs(d)) for a, b, c, d in zip(sc["chi2"], sn["chi2"], sc["robust_chi2"], sn["robust_chi2"])]
    agree = 0
    while agree < len(rel) and rel[agree] <= 1e-6:
        agree += 1
    print(f"{name} init=odom: C++ oracle {tc:.1f}s, numpy oracle {tn:.1f}s; iterates 0..{agree - 1} agree to 1e-6 (of {iters + 1})")
    for k, r in enumerate(rel):
        print(f"  it {k:2d}: chi2 {sc['chi2'][k]:.9g} robust {sc['robust_chi2'][k]:.9g}  rel diff between the oracles {r:.2e}")
    out = os.path.join(ROOT, "tests", "golden", f"{name}_odom_direct.npz")
    np.savez_compressed(out, config=name, init="odom", iters=iters, V=g.V, E=g.E, chi2=np.array(sc["chi2"]),
                        robust_chi2=np.array(sc["robust_chi2"]), np_chi2=np.array(sn["chi2"]), np_robust_chi2=np.array(sn["robust_chi2"]),
                        agree=agree, oracle_rel_diff=np.array(rel), digest=graph_digest(g))
    print("wrote", out)


if __name__ == "__main__":
    main()
