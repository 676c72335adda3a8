#!/usr/bin/env python3
"""Generate the large-config golden fixture: direct-solver oracle GN x20 on a synth config.

Usage: python scripts/make_golden_large.py C4 [iters] [direct|pcg] [V E]
  C4  direct   about 15 minutes of one core
  C4r pcg      about 15 minutes (5 % random closures: a sparse direct factorisation fills in
               catastrophically, SURVEY.md section 8(d), so the oracle's own block-Jacobi PCG at
               1e-10 is the reference: "PCG-vs-PCG")
  C5 pcg V E   a C5-shaped graph (same generator arguments, smaller V / E) the oracle can finish
Writes tests/golden/<config>[_V]_<solver>.npz: chi2 / robust chi2 per iteration, every 50th final
pose, and a digest of the generated graph (to detect generator drift).
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402
from sparse_gslam_amd import synth  # noqa: E402


def graph_digest(g):
    h = hashlib.sha256()
    for a in g.arrays():
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    solver = sys.argv[3] if len(sys.argv) > 3 else "direct"
    over = dict(V=int(sys.argv[4]), E=int(sys.argv[5])) if len(sys.argv) > 5 else {}
    g = synth.config(name, **over)
    t = time.time()
    P, st = c_oracle.gauss_newton(*g.arrays(), iters=iters, solver=solver, pcg_tol=1e-10, pcg_maxit=500000)
    print(f"{name}: oracle GN x{iters} {solver} took {time.time() - t:.1f}s; chi2 {st['chi2'][0]:.9g} -> {st['chi2'][-1]:.12g}"
          f"; pcg iterations {[int(k) for k in st['pcg_iters']]}")
    tag = name + (f"_{over['V']}" if over else "")
    out = os.path.join(ROOT, "tests", "golden", f"{tag}_{solver}.npz")
    np.savez_compressed(out, config=name, iters=iters, V=g.V, E=g.E, solver=solver, pcg_iters=np.array(st["pcg_iters"]),
                        chi2=np.array(st["chi2"]),
                        robust_chi2=np.array(st["robust_chi2"]), poses_stride50=P[::50].copy(),
                        pose_sum=P.sum(axis=0), pose_abs_sum=np.abs(P).sum(axis=0),
                        digest=graph_digest(g), seconds=np.array(st["seconds"]))
    print("wrote", out)


if __name__ == "__main__":
    main()
