#!/usr/bin/env python3
"""Probe: the odometry chain as a DIRECT component of the preconditioner on BASELINE.md's literal
dead-reckoned start.  H = T + W: T the block-tridiagonal part of the Hessian in trajectory (= hessian) order -- the odometry
chain plus every diagonal block --, W the closures' off-diagonal blocks.  Counts PCG iterations to `tol` for
  jac      block-Jacobi
  chain    T^-1 alone (banded Cholesky)
  amg      the library's multigrid cycle (sgo_precondition; GPU)
  add      T^-1 r + M_amg r                         (additive)
  mult     chain -> amg -> chain, symmetric          (two extra Hessian products)
  cheap    T^-1 r, then M_amg (r - H T^-1 r)         (non-symmetric, flexible CG; one extra product)
Usage: python scripts/chain_precond_probe.py [config=C4] [init=odom] [tol=1e-8] [gn_steps=1] [which=jac,chain,amg,add,mult,cheap]
Without a GPU only jac / chain run."""
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_oracle  # noqa: E402
from sparse_gslam_amd import synth  # noqa: E402


def band_of(H, kd=5):
    n = H.shape[0]
    ab = np.zeros((kd + 1, n))
    for k in range(kd + 1):
        d = H.diagonal(-k)
        ab[k, : n - k] = d
    return ab


def fpcg(matvec, prec, b, tol, maxit, flexible=True):
    x = np.zeros_like(b)
    r = b.copy()
    z = prec(r)
    p = z.copy()
    rz = r @ z
    bn = np.sqrt(b @ b)
    it = 0
    hist = []
    while it < maxit:
        q = matvec(p)
        alpha = rz / (p @ q)
        x += alpha * p
        r_old = r.copy() if flexible else None
        r -= alpha * q
        it += 1
        rn = np.sqrt(r @ r) / bn
        hist.append(rn)
        if rn <= tol:
            break
        z = prec(r)
        rz_new = r @ z
        beta = (z @ (r - r_old)) / rz if flexible else rz_new / rz
        p = z + beta * p
        rz = rz_new
    return x, it, hist


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    init = sys.argv[2] if len(sys.argv) > 2 else "odom"
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
    gn_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    which = (sys.argv[5] if len(sys.argv) > 5 else "jac,chain,amg,add,mult,cheap").split(",")
    g = synth.config(name, init=init)
    poses = g.poses.copy()
    opt = None
    try:
        from sparse_gslam_amd import capi
        opt = capi.Optimizer(0, pcg_tol=tol)
        opt.set_graph(*g.arrays())
        print("solver:", opt.solver_description()[:400])
    except Exception as e:   # no GPU here
        print("no GPU context:", e)
        which = [w for w in which if w in ("jac", "chain")]
    for step in range(gn_steps):
        t = time.time()
        H, b, c2, rc2 = np_oracle.linearize(poses, g.fixed, g.ei, g.ej, g.meas, g.info, g.phi)
        H = H.tocsr()
        n3 = b.size
        ab = band_of(H)
        cb = sla.cholesky_banded(ab, lower=True)
        print(f"GN step {step}: chi2 {c2:.6g} robust {rc2:.6g}; system built in {time.time() - t:.1f}s")
        if opt is not None:
            opt.set_poses(poses)
            gb, _, _, _ = opt.linearize()
            assert np.abs(gb.ravel() - b).max() <= 1e-9 * np.abs(b).max()
        D = np.stack([np.stack([H.diagonal(c - r)[r::3][: n3 // 3] if c >= r else H.diagonal(c - r)[c::3][: n3 // 3] for c in range(3)], axis=-1)
                      for r in range(3)], axis=-2)   # (n,3,3) diagonal blocks
        Dinv = np.linalg.inv(D)
        mv = lambda v: H @ v   # noqa: E731
        chain = lambda r: sla.cho_solve_banded((cb, True), r)   # noqa: E731
        jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(-1, 3)).ravel()   # noqa: E731
        amg = (lambda r: opt.precondition(r.reshape(-1, 3)).ravel()) if opt is not None else None

        def mult(r):
            z = chain(r)
            z = z + amg(r - mv(z))
            return z + chain(r - mv(z))

        def cheap(r):
            z = chain(r)
            return z + amg(r - mv(z))

        precs = dict(jac=jac, chain=chain, amg=amg, add=lambda r: chain(r) + amg(r), mult=mult, cheap=cheap)
        xs = {}
        for w in which:
            t = time.time()
            x, it, hist = fpcg(mv, precs[w], b, tol, 3000 if w != "jac" else 300, flexible=w not in ("jac", "chain"))
            xs[w] = x
            marks = [hist[min(k, len(hist) - 1)] for k in (9, 19, 49, 99, 199)]
            print(f"  {w:6s}: {it:5d} iterations to {hist[-1]:.2e}  ({time.time() - t:.1f}s; relres after 10/20/50/100/200: "
                  + " ".join(f"{m:.1e}" for m in marks) + ")", flush=True)
        if gn_steps > 1:
            x = xs.get("amg", xs.get("chain"))
            poses = np_oracle.oplus(poses, g.fixed, x)
    if opt is not None:
        opt.close()


if __name__ == "__main__":
    main()
