"""numpy prototype: what the resident multigrid hierarchy loses by not seeing an incremental update's appended closures (DESIGN.md
section 5b: + 4-5 PCG iterations per solve on the C4-sized session), and how much of it comes back when only the COARSEST operator
receives the Galerkin image of the update, W^T (H_eff - H_res) W with W = P_0 P_1 ... (a dense rank-3|T| correction of a 660 x 660
matrix before it is inverted).

  python scripts/proto/overlay_coarse.py V E steps chain

(a) hierarchy of the resident graph, PCG on the Schur complement H_eff      -- what the product does
(b) the same, coarsest operator corrected
(c) hierarchy built from H_eff itself                                        -- a fresh set-up
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
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
chain = int(sys.argv[4]) if len(sys.argv) > 4 else 25


def pcg(Hop, levels, b, maxit=2000):
    x = np.zeros_like(b); r = b.copy(); z = cyc(levels, 0, r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b); it = 0
    while it < maxit:
        q = Hop @ p; a = rz / (p @ q); x += a * p; r -= a * q; it += 1
        if np.linalg.norm(r) <= 1e-8 * bn:
            break
        z = cyc(levels, 0, r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn
    return it


base, app, g = synth.append_session(V, E, steps, chain, 4)
odom_meas = g.meas[: g.V - 1]
arrs = [base.ei, base.ej, base.meas, base.info, base.phi]
P = base.poses.copy()
levels_res = None
for k, a in enumerate(app):
    arrs = [np.concatenate([x, a[n]]) for x, n in zip(arrs, ("ei", "ej", "meas", "info", "phi"))]
    Vk = a["V"]
    P0 = np.empty((Vk, 3)); P0[: P.shape[0]] = P
    synth.chain_init(P0, odom_meas, P.shape[0], Vk - 1)
    fixed = np.zeros(Vk, dtype=bool); fixed[0] = True
    nres_e = base.E
    # resident operator: the base graph's edges only; full operator: all edges
    Hres, bres, _, _ = np_oracle.linearize(P0[: base.V], fixed[: base.V], *[x[:nres_e] for x in arrs])
    Hfull, bfull, _, _ = np_oracle.linearize(P0, fixed, *arrs)
    nR = 3 * (base.V - 1)
    Hfull = Hfull.tocsr()
    HRR, HRN, HNN = Hfull[:nR, :nR], Hfull[:nR, nR:], Hfull[nR:, nR:]
    lu = spla.splu(HNN.tocsc())
    tc = np.unique(HRN.nonzero()[0])                 # the touched rows' scalar indices: H_RN is zero elsewhere
    Xt = lu.solve(HRN[tc].T.toarray())               # H_NN^-1 H_NT
    St = HRN[tc] @ Xt                                # |T| x |T| dense Schur term
    S = sp.coo_matrix((St.ravel(), (np.repeat(tc, len(tc)), np.tile(tc, len(tc)))), shape=(nR, nR)).tocsr()
    S.data[np.abs(S.data) < 1e-300] = 0; S.eliminate_zeros()
    Heff = (HRR - S).tocsr()
    beff = bfull[:nR] - HRN @ lu.solve(bfull[nR:])
    pos = P0[1: base.V, :2]
    if levels_res is None:
        levels_res = build(Hres.tocsr(), pos)        # made once, as the resident hierarchy is (values refreshed below)
    lev_a = build(Hres.tocsr(), pos, reuse=levels_res)
    it_a = pcg(Heff, lev_a, beff)
    # (b): coarsest operator + W^T (Heff - Hres) W
    W = sp.identity(nR, format="csr")
    for L in lev_a[:-1]:
        W = W @ L["P"]
    dH = (Heff - Hres.tocsr()).tocsr()
    Ac = lev_a[-1]["A"] + (W.T @ dH @ W)
    lev_b = [dict(L) for L in lev_a]
    lev_b[-1] = dict(lev_a[-1]); lev_b[-1]["lu"] = spla.splu(sp.csc_matrix(Ac))
    it_b = pcg(Heff, lev_b, beff)
    lev_c = build(Heff, pos)
    it_c = pcg(Heff, lev_c, beff)
    T = np.unique(dH.nonzero()[0] // 3)
    print(f"update {k}: {len(T)} touched rows; PCG iterations  (a) blind {it_a}   (b) coarsest corrected {it_b}   (c) fresh hierarchy {it_c}"
          f"   levels {[L['n'] for L in lev_a]}", flush=True)
    # the session goes on from here without optimising (the prototype is about one linear solve per update)
    P = P0
