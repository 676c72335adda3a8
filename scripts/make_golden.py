#!/usr/bin/env python3
"""Generate the small golden fixtures under tests/golden/ from the numpy/scipy restatement.

The reference has no tests or golden vectors for the optimiser path and its
optimiser (g2o) cannot be built or imported here, so these vectors come from oracle/np_oracle.py
(exact sparse direct solve, fp64) and are cross-checked against the independent C++ restatement
when they are generated.  Each fixture is self-contained: graph arrays + expected chi2 per
iteration + expected final poses.

    python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, np_oracle  # noqa: E402
from sparse_gslam_amd import synth  # noqa: E402

CASES = {
    # name: generator kwargs
    "tiny_full": dict(V=120, E=220, seed=11, info_mode="full"),
    "tiny_odom": dict(V=60, E=90, seed=12, info_mode="full", init="odom", phi=10.0),
    "tiny_random": dict(V=200, E=500, seed=13, p_random=0.2, info_mode="diag"),
    "tiny_chain": dict(V=50, E=49, seed=14, info_mode="full", tail=10),   # odometry only: chi2 -> 0
}


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, kw in CASES.items():
        g = synth.manhattan(**kw)
        trace = []
        P, st = np_oracle.gauss_newton(*g.arrays(), iters=20, trace=trace)
        Pc, sc = c_oracle.gauss_newton(*g.arrays(), iters=20)
        rel = max(abs(a - b) / max(abs(a), 1e-30) for a, b in zip(st["chi2"], sc["chi2"]) if a > 1e-12)  \
            if max(st["chi2"]) > 1e-12 else 0.0
        dp = np.abs(P - Pc).max()
        assert rel < 1e-7 and dp < 1e-7, (name, rel, dp)
        H, b, _, _ = np_oracle.linearize(*g.arrays())
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), poses=g.poses, fixed=g.fixed, ei=g.ei, ej=g.ej,
            meas=g.meas, info=g.info, phi=g.phi, chi2=np.array(st["chi2"]),
            robust_chi2=np.array(st["robust_chi2"]), final_poses=P, poses_iter1=trace[0], b0=b,
            e2_0=np_oracle.chi2(*[g.arrays()[k] for k in (0, 2, 3, 4, 5, 6)])[2])
        print(f"{name}: V={g.V} E={g.E} chi2 {st['chi2'][0]:.6g} -> {st['chi2'][-1]:.6g}; "
              f"np vs c++: rel chi2 {rel:.1e}, poses {dp:.1e}")


if __name__ == "__main__":
    main()
