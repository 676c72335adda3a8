"""numpy prototype: what the coarse operators of a Gauss-Newton iteration may be made from, short of a full refresh.

Along optimize(20) of a graph from the bench's incremental start (the CPU oracle's iterates), at every iteration k >= 1 the
hierarchy's aggregates are those of iteration 0 and PCG (1e-8) on the CURRENT Hessian is preconditioned by a cycle whose level-0
smoother uses the current Hessian and whose coarse operators are

  fresh      P and P^T A P from the current Hessian                                  -- what a refresh costs 1.2 ms for on C4
  kept       everything from iteration k-1                                            -- the lagged refresh's kept solve
  frozen P   P from iteration k-1, P^T A P from the current Hessian
  mixed tau  P from iteration k-1, P^T A_mix P with A_mix = the edges whose DCS weight moved by more than tau (relative) or whose
             end poses turned by more than tau radians taken at the current poses, all other edges as they were at iteration k-1:
             the incremental Galerkin update A_c += P^T dA P over the changed slots only

  tentative  P from iteration k-1, A_1 += T^T (H_k - H_{k-1}) T with the TENTATIVE prolongator T: one product per level-0 block
  chained    the same carried forward from the last full refresh (every FULL-th iteration), transfers frozen there

  delta chained  transfers frozen at iteration 0; an edge re-enters the coarse operators at the current poses when its weight or an
             end angle has moved by more than CT since it was last included -- the delta refresh as it would run on the device

  python scripts/proto/incremental_galerkin.py V E [iters] [FULL] [CT]
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import np_oracle  # noqa: E402
from sparse_gslam_amd import synth  # noqa: E402
import fsa_lib  # noqa: E402
from fsa_lib import build, cyc  # noqa: E402

fsa_lib.np, fsa_lib.sp, fsa_lib.spla = np, sp, spla
V, E = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
omega = fsa_lib.omega


def dinv_of(A):
    n = A.shape[0] // 3
    B = A.tobsr(blocksize=(3, 3)); B.sort_indices()
    rows = np.repeat(np.arange(n), np.diff(B.indptr)); dm = rows == B.indices
    D = np.zeros((n, 3, 3)); D[rows[dm]] = B.data[dm]
    return sp.bsr_matrix((np.linalg.inv(0.5 * (D + D.transpose(0, 2, 1))), np.arange(n), np.arange(n + 1)), shape=A.shape).tocsr()


def with_P(old, H_true, H_coarse_from):
    """levels with the transfers of `old`, level 0 = H_true (smoother, residual), coarse operators from H_coarse_from."""
    out = []
    A = H_coarse_from.tocsr()
    for l, L in enumerate(old):
        lev = dict(n=L["n"])
        lev["A"] = H_true.tocsr() if l == 0 else A
        lev["Dinv"] = dinv_of(lev["A"])
        if "P" in L:
            lev["P"] = L["P"]
            A = (L["P"].T @ A @ L["P"]).tocsr()
        else:
            lev["lu"] = spla.splu(sp.csc_matrix(A))
        out.append(lev)
    return out


def with_A1(old, H_true, A1):
    """transfers of `old`; level 0 = H_true; level 1 = A1 as given; levels below from it with the old transfers."""
    out = []
    A = None
    for l, L in enumerate(old):
        lev = dict(n=L["n"])
        lev["A"] = H_true.tocsr() if l == 0 else (A1.tocsr() if l == 1 else A)
        lev["Dinv"] = dinv_of(lev["A"])
        if "P" in L:
            lev["P"] = L["P"]
            if l >= 1:
                A = (L["P"].T @ lev["A"] @ L["P"]).tocsr()
        else:
            lev["lu"] = spla.splu(sp.csc_matrix(lev["A"]))
        out.append(lev)
    return out


def tentative(levels0, pos):
    """the tentative (rigid-body, piecewise constant) prolongator of level 0 for the aggregates of levels0 at positions pos"""
    agg, nc = levels0[0]["agg"], levels0[0]["nc"]
    n = levels0[0]["n"]
    cent = np.zeros((nc, 2)); np.add.at(cent, agg, pos); cent /= np.bincount(agg, minlength=nc)[:, None]
    return sp.bsr_matrix((fsa_lib.Tm(pos - cent[agg]), agg, np.arange(n + 1)), shape=(3 * n, 3 * nc)).tocsr()


def kept(old, H_true):
    out = [dict(L) for L in old]
    out[0] = dict(old[0]); out[0]["A"] = H_true.tocsr(); out[0]["Dinv"] = dinv_of(H_true)
    return out


def pcg(Hop, levels, b, maxit=1500):
    x = np.zeros_like(b); r = b.copy(); z = cyc(levels, 0, r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b); it = 0
    while it < maxit:
        q = Hop @ p; a = rz / (p @ q); x += a * p; r -= a * q; it += 1
        if np.linalg.norm(r) <= 1e-8 * bn:
            break
        z = cyc(levels, 0, r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn
    return it, x


g = synth.manhattan(V, E, seed=4)
fixed, ei, ej, meas, info, phi = g.fixed, g.ei, g.ej, g.meas, g.info, g.phi
poses = g.poses.copy()
free = np.flatnonzero(~fixed)
H0, b0, _, _ = np_oracle.linearize(poses, fixed, ei, ej, meas, info, phi)
agg_levels = build(H0.tocsr(), poses[free, :2])
prev_levels, prev_poses, prev_w, prev_H = None, None, None, None
FULL = int(sys.argv[4]) if len(sys.argv) > 4 else 100
chain_levels = chain_A1 = chain_pos = None
CT = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
hist_poses = []
print(f"V={V} E={E}; levels {[L['n'] for L in agg_levels]}")
for k in range(iters):
    H, b, c2, rc2 = np_oracle.linearize(poses, fixed, ei, ej, meas, info, phi)
    H = H.tocsr()
    _, _, e2 = np_oracle.chi2(poses, ei, ej, meas, info, phi)
    _, w = np_oracle.dcs_rho(e2, phi)
    fresh = build(H, poses[free, :2], reuse=agg_levels)
    it_f, dx = pcg(H, fresh, b)
    line = f"it {k:2d}  fresh {it_f:3d}"
    if prev_levels is not None:
        it_k, _ = pcg(H, kept(prev_levels, H), b)
        it_p, _ = pcg(H, with_P(prev_levels, H, H), b)
        line += f"   kept {it_k:3d}   frozen P {it_p:3d}"
        # the CHEAP update: A_1 += T^T (H - H_prev) T with the tentative prolongator T (one pass over the level-0 blocks with a
        # Galerkin map that has one product per block -- k_galerkin, 71 us on C4 -- instead of A P and P^T A P, 510 us)
        T = tentative(agg_levels, prev_poses[free, :2])
        A1 = (prev_levels[1]["A"] + T.T @ (H - prev_H) @ T).tocsr()
        it_t, _ = pcg(H, with_A1(prev_levels, H, A1), b)
        line += f"   tentative update {it_t:3d}"
        # ... and the same CHAINED: transfers and lower-level transfers frozen at the last full refresh (iteration 0 / every FULL-th),
        # A_1 carried forward by tentative updates only
        if k % FULL == 0:
            chain_levels, chain_A1 = fresh, fresh[1]["A"]
            line += "   chained   full"
        else:
            Tc = tentative(agg_levels, chain_pos)
            chain_A1 = (chain_A1 + Tc.T @ (H - prev_H) @ Tc).tocsr()
            it_c, _ = pcg(H, with_A1(chain_levels, H, chain_A1), b)
            line += f"   chained {it_c:3d}"
        # the delta refresh as it would run: transfers frozen at iteration 0, H_mix carried forward -- an edge is re-included at the
        # current poses when its DCS weight has moved by more than CT (relative) or an end pose has turned by more than CT since the
        # edge was last included
        if k >= 1:
            w_inc = np.array([hist_w[last_inc[e]][e] for e in range(len(ei))]) if False else inc_w
            dth_inc = np.maximum(np.abs(np_oracle.normalize_theta(poses[ei, 2] - inc_th_i)), np.abs(np_oracle.normalize_theta(poses[ej, 2] - inc_th_j)))
            F = (np.abs(w - inc_w) > CT * np.maximum(w, inc_w)) | (dth_inc > CT)
            for j in np.unique(last_inc[F]):
                sel = F & (last_inc == j)
                Hj, _, _, _ = np_oracle.linearize(hist_poses[j], fixed, ei[sel], ej[sel], meas[sel], info[sel], phi[sel])
                H_mix = H_mix - Hj
            if F.any():
                Hn, _, _, _ = np_oracle.linearize(poses, fixed, ei[F], ej[F], meas[F], info[F], phi[F])
                H_mix = (H_mix + Hn).tocsr()
            last_inc[F] = k; inc_w[F] = w[F]; inc_th_i[F] = poses[ei[F], 2]; inc_th_j[F] = poses[ej[F], 2]
            it_d, _ = pcg(H, with_P(agg_levels_fresh0, H, H_mix), b)
            line += f"   delta chained ({CT}) {it_d:3d} ({int(F.sum())} edges)"
        dth = np.abs(np_oracle.normalize_theta(poses[:, 2] - prev_poses[:, 2]))
        for tau in (0.2, 0.05, 0.01):
            ch = (np.abs(w - prev_w) > tau * np.maximum(w, prev_w)) | (dth[ei] > tau) | (dth[ej] > tau)
            Hc, _, _, _ = np_oracle.linearize(poses, fixed, ei[ch], ej[ch], meas[ch], info[ch], phi[ch]) if ch.any() else (sp.csr_matrix(H.shape), 0, 0, 0)
            Hu, _, _, _ = np_oracle.linearize(prev_poses, fixed, ei[~ch], ej[~ch], meas[~ch], info[~ch], phi[~ch])
            it_m, _ = pcg(H, with_P(prev_levels, H, (Hc + Hu).tocsr()), b)
            line += f"   mixed {tau}: {it_m:3d} ({int(ch.sum())} edges)"
    print(line, flush=True)
    hist_poses.append(poses.copy())
    if k == 0:
        agg_levels_fresh0, H_mix = fresh, H.copy()
        last_inc = np.zeros(len(ei), dtype=int); inc_w = w.copy(); inc_th_i = poses[ei, 2].copy(); inc_th_j = poses[ej, 2].copy()
    if prev_levels is None or k % FULL == 0:
        chain_levels, chain_A1, chain_pos = fresh, fresh[1]["A"], poses[free, :2].copy()
    prev_levels, prev_poses, prev_w, prev_H = fresh, poses.copy(), w, H
    poses = np_oracle.oplus(poses, fixed, dx)
