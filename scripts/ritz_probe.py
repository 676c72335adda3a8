#!/usr/bin/env python3
"""Ritz values of the preconditioned operator M^-1 H of a solve (from the PCG recurrence's alpha / beta: the Lanczos
tridiagonal T has diagonal 1/alpha_j + beta_{j-1}/alpha_{j-1} and off-diagonal sqrt(beta_j)/alpha_j): are the iteration
counts set by a few isolated small eigenvalues (deflation / recycling would pay) or by the bulk of the spectrum?
Usage: SGO_LANCZOS=1 python scripts/ritz_probe.py [config] [gn_iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SGO_LANCZOS"] = "1"
from sparse_gslam_amd import capi, synth  # noqa: E402


def ritz(alpha, beta):
    m = len(alpha)
    T = np.zeros((m, m))
    for j in range(m):
        T[j, j] = 1.0 / alpha[j] + (beta[j - 1] / alpha[j - 1] if j > 0 else 0.0)
        if j + 1 < m:
            T[j, j + 1] = T[j + 1, j] = np.sqrt(max(beta[j], 0.0)) / alpha[j]
    return np.linalg.eigvalsh(T)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    g = synth.config(cfg)
    with capi.Optimizer(0, pcg_warm_start=0) as o:   # (cold starts: the Krylov space then starts from b)
        o.set_graph(*g.arrays())
        for it in range(iters):
            d, st = o.optimize(1)
            a, b = o.lanczos()
            ev = ritz(a, b)
            print(f"GN {it}: {len(a)} PCG iterations; Ritz values min {ev[0]:.4f} max {ev[-1]:.4f}; smallest 8: "
                  + " ".join(f"{v:.4f}" for v in ev[:8]) + " | largest 4: " + " ".join(f"{v:.3f}" for v in ev[-4:]), flush=True)


if __name__ == "__main__":
    main()
