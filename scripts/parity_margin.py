#!/usr/bin/env python3
"""Parity margin vs PCG tolerance: max relative chi2 error per GN iterate against the committed goldens /
the CPU oracle, for several pcg_tol values (decides the default tolerance; DESIGN.md section 3, NOTES.md section 7)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle  # noqa: E402
from sparse_gslam_amd import capi, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
tols = [float(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1e-8", "1e-7", "3e-7"])]


def run(g, ref_chi2, ref_poses, tol, iters=20, stride=None):
    with capi.Optimizer(0, pcg_tol=tol) as o:
        o.set_graph(*g.arrays())
        done, st = o.optimize(iters)
        P = o.get_poses()
    rel = max(abs(a - b) / b for a, b in zip(st["chi2"], ref_chi2))
    perr = np.abs((P[::stride] if stride else P) - ref_poses).max()
    return rel, perr, float(np.mean(st["pcg_iters"])), 1e3 * float(np.median(st["seconds"]))


cases = []
for name, file, kw in (("C4", "C4_direct.npz", {}), ("C4r", "C4r_pcg.npz", {}), ("C5s", "C5_50000_pcg.npz", None)):
    f = np.load(os.path.join(GOLD, file))
    g = synth.config("C5", V=int(f["V"]), E=int(f["E"])) if kw is None else synth.config(name)
    cases.append((name, g, f["chi2"], f["poses_stride50"], 50))
g = synth.config("C2", info_mode="full")
oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=20)
cases.append(("C2 full", g, ost["chi2"], oP, None))
rng = np.random.default_rng(2024)
for V, dens in [(450, 3.0), (3000, 6.0), (2000, 3.0), (1200, 1.5), (3000, 1.05), (8000, 4.0)]:
    g = synth.manhattan(V=V, E=int(dens * V), seed=int(rng.integers(1, 10**6)), p_random=float(rng.choice([0.0, 0.05, 0.3])),
                        info_mode=str(rng.choice(["diag", "full"])), phi=float(rng.choice([1.0, 10.0])))
    oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=20)
    cases.append((f"rand V={V} d={dens}", g, ost["chi2"], oP, None))
for name, g, rc, rp, stride in cases:
    line = f"{name:18s}"
    for tol in tols:
        rel, perr, k, ms = run(g, rc, rp, tol, stride=stride)
        line += f" | tol {tol:.0e}: chi2 rel {rel:.1e} poses {perr:.1e} pcg {k:.1f} {ms:.2f} ms"
    print(line, flush=True)
