"""Host-only checks of the multifrontal path's elimination plan (sgo_mfront_plan: nested dissection with minimum-vertex-cover
separators, sgo_mfront.h) -- no GPU: the plan covers every free pose exactly once, its fill agrees with an independent symbolic
factorisation in the same order, and the refusals say why."""
import numpy as np
import scipy.sparse as sp

from sparse_gslam_amd import capi, synth


def _free_adjacency(g):
    free = ~g.fixed.astype(bool)
    deg = np.bincount(np.concatenate([g.ei, g.ej]), minlength=g.V)
    ids = np.flatnonzero(free & (deg > 0))
    return ids


def _symbolic_fill(order_vertices, g):
    """nnz(L) in 3x3 blocks (incl. the diagonal) of the Cholesky factor in the given elimination order: plain elimination game."""
    pos = -np.ones(g.V, dtype=np.int64)
    pos[order_vertices] = np.arange(len(order_vertices))
    adj = [set() for _ in order_vertices]
    for a, b in zip(pos[g.ei], pos[g.ej]):
        if a >= 0 and b >= 0:
            adj[min(a, b)].add(int(max(a, b)))
    nnz = 0
    for k in range(len(adj)):
        up = sorted(adj[k])
        nnz += 1 + len(up)
        if up:
            p = up[0]
            adj[p].update(up[1:])
    return nnz


def test_plan_is_a_permutation_and_its_fronts_hold_the_fill():
    g = synth.config("C3s")
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej)
    ids = _free_adjacency(g)
    assert r["qualifies"] and r["n"] == len(ids) == 5488
    assert sorted(r["elim_vertex"].tolist()) == ids.tolist()
    f = r["front_of_elim"]
    assert (np.diff(f) >= 0).all() and f[0] == 0 and f[-1] == r["fronts"] - 1       # fronts own consecutive positions, children first
    assert r["levels"] <= 12 and r["max_dim"] <= 400 and r["crit_flops"] < 40e6 < r["flops"]
    # the fronts' storage (lower triangles of own x (own + boundary) blocks) is what the elimination game produces in this
    # order, up to the dense treatment of every front's own block (an upper bound, and within 30 % of the exact fill)
    exact = _symbolic_fill(r["elim_vertex"], g)
    own = np.bincount(f)
    assert exact >= r["n"]
    dense_own = int((own * (own + 1) // 2).sum())
    assert dense_own <= exact * 1.3 + 1 and exact < 60 * r["n"]


def test_plan_refusals_say_why():
    g = synth.config("C2")
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej)
    assert not r["qualifies"] and "4.00 edges per free pose" in r["why"] and r["n"] == 9999
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej, 0, 1e9)          # an explicit budget: analysed whatever the density
    assert r["qualifies"] and r["max_dim"] > 400 and r["crit_flops"] > 80e6
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej, 0, 100.0)
    assert not r["qualifies"] and "Mflop on the critical path" in r["why"]
    g = synth.manhattan(16000, 21000, seed=7, info_mode="full", phi=0.75)   # refused from its three top separators alone
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej)
    assert not r["qualifies"] and "already in its two top levels" in r["why"], r["why"]


def test_both_row_orders_are_tried_and_tiny_graphs_are_one_front():
    g = synth.manhattan(1000, 1100, seed=1, info_mode="full")           # a chain with few closures: the id order is as good
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej)
    assert r["qualifies"] and r["order_kind"] in (0, 1) and r["levels"] >= 4
    poses = g.poses.copy()
    poses[:, :2] = 0.0                                                   # no geometry: only the id order is left
    r0 = capi.mfront_plan(poses, g.fixed, g.ei, g.ej)
    assert r0["qualifies"] and r0["order_kind"] == 1
    g = synth.manhattan(20, 25, seed=2, info_mode="full")
    r = capi.mfront_plan(g.poses, g.fixed, g.ei, g.ej)
    assert r["qualifies"] and r["fronts"] == 1 and r["levels"] == 1 and r["max_own"] == 19 and r["max_bnd"] == 0
