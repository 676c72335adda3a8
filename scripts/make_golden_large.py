#!/usr/bin/env python3
"""Generate the large-config golden fixture: direct-solver oracle GN x20 on a synth config.

Usage: python scripts/make_golden_large.py C4   (about 15 minutes of one core for C4)
Writes tests/golden/<config>_direct.npz: chi2 / robust chi2 per iteration, every 50th final
pose, and checksums of the generated graph (to detect generator drift).
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
    g = synth.config(name)
    t = time.time()
    P, st = c_oracle.gauss_newton(*g.arrays(), iters=iters, solver="direct")
    print(f"{name}: oracle GN x{iters} direct took {time.time() - t:.1f}s; chi2 {st['chi2'][0]:.9g} -> {st['chi2'][-1]:.12g}")
    out = os.path.join(ROOT, "tests", "golden", f"{name}_direct.npz")
    np.savez_compressed(out, config=name, iters=iters, chi2=np.array(st["chi2"]),
                        robust_chi2=np.array(st["robust_chi2"]), poses_stride50=P[::50].copy(),
                        pose_sum=P.sum(axis=0), pose_abs_sum=np.abs(P).sum(axis=0),
                        digest=graph_digest(g), seconds=np.array(st["seconds"]))
    print("wrote", out)


if __name__ == "__main__":
    main()
