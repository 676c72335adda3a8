"""The mid-size path (sgo_mfront.h): sgo_optimize_gn through a multifrontal sparse Cholesky factorisation, one launch per
level of a nested-dissection elimination tree -- the graphs between the single-launch direct path and the multigrid PCG
(the reference's largest: mit-killian's 5 489 poses / 7 629 edges, re-optimised after every closure through g2o's sparse
Cholesky: src/sparse_gslam/src/submap_loop_closer.cpp:286-287, src/sparse_gslam/src/graphs.cpp:19).

Parity against the CPU oracle's direct solver on the same inputs (fp64; two direct factorisations differ by rounding,
amplified by the Hessian's conditioning: chi2 of every iterate within 1e-8 relative on these sizes where BASELINE.json
asks for 1e-6), against the committed C3s fixture, against the multigrid PCG path of the same library, and the edge cases
of the reference's call sites.  direct_rows=1 makes the single-launch path refuse graphs it would otherwise take.
"""
import hashlib
import os

import numpy as np
import pytest

from sparse_gslam_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _oracle():
    from oracle import c_oracle
    return c_oracle


def chain_graph(V, closures, seed=1, **kw):
    kw.setdefault("info_mode", "full")
    kw.setdefault("init", "odom")
    kw.setdefault("phi", 10.0)
    return synth.manhattan(V, V - 1 + closures, seed=seed, **kw)


def run(args, iters=20, **opts):
    opts.setdefault("direct_rows", 1)
    with capi.Optimizer(0, **opts) as o:
        o.set_graph(*args)
        desc = o.solver_description()
        done, st = o.optimize(iters)
        return desc, done, st, o.get_poses()


def test_c3s_takes_the_multifrontal_path_and_matches_the_golden():
    """The stand-in of configs[2]'s graph (5 489 poses / 7 629 edges, DCS 0.75, full information) through the default
    path against the sparse-direct-solver oracle's fixture: every iterate's chi2 within BASELINE.json's 1e-6."""
    f = np.load(os.path.join(GOLDEN, "C3s_direct.npz"))
    g = synth.config("C3s")
    h = hashlib.sha256()
    for a in g.arrays():
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(f["digest"]), "generator drift: the graph is not the one the fixture was made from"
    with capi.Optimizer(0) as opt:
        opt.set_graph(*g.arrays())
        d = opt.solver_description()
        assert d.startswith("multifrontal_cholesky") and "direct path not used" in d, d
        done, st = opt.optimize(20)
        P = opt.get_poses()
    assert done == 20 and st["pcg_iters"] == [0] * 20
    for k in range(21):
        assert abs(st["chi2"][k] - f["chi2"][k]) <= 1e-6 * f["chi2"][k], k
        assert abs(st["robust_chi2"][k] - f["robust_chi2"][k]) <= 1e-6 * f["robust_chi2"][k], k
    assert np.abs(P[::50] - f["poses_stride50"]).max() <= 1e-5


@pytest.mark.parametrize("V,closures,seed", [(3, 0, 1), (4, 1, 2), (17, 2, 3), (33, 3, 4), (64, 5, 5), (65, 20, 6), (300, 60, 7),
                                             (1000, 300, 8), (2000, 900, 9)])
def test_graphs_of_many_sizes_match_the_oracle(V, closures, seed):
    """One leaf front (V <= 32), two levels, many levels; few closures and a closure every two or three poses."""
    g = chain_graph(V, closures, seed)
    desc, done, st, P = run(g.arrays())
    assert desc.startswith("multifrontal_cholesky"), desc
    oP, ost = _oracle().gauss_newton(*g.arrays(), iters=20)
    assert done == ost["iters_done"] == 20
    for k in range(21):
        assert abs(st["chi2"][k] - ost["chi2"][k]) <= 1e-8 * ost["chi2"][k] + 1e-18, k
        assert abs(st["robust_chi2"][k] - ost["robust_chi2"][k]) <= 1e-8 * ost["robust_chi2"][k] + 1e-18, k
    assert np.abs(P - oP).max() < 1e-7


@pytest.mark.parametrize("info_mode,phi,init", [("diag", 1.0, "incremental"), ("full", 0.75, "incremental"), ("full", -1.0, "odom")])
def test_information_and_kernel_variants_match_the_oracle(info_mode, phi, init):
    """Diagonal and full information matrices, a DCS kernel that switches closures off, no kernel at all."""
    g = synth.manhattan(1500, 2600, seed=31, info_mode=info_mode, phi=phi, init=init)
    desc, done, st, P = run(g.arrays())
    assert desc.startswith("multifrontal_cholesky"), desc
    oP, ost = _oracle().gauss_newton(*g.arrays(), iters=20)
    assert done == ost["iters_done"] == 20
    rel = np.abs(np.array(st["chi2"]) - np.array(ost["chi2"])) / np.array(ost["chi2"])
    rrel = np.abs(np.array(st["robust_chi2"]) - np.array(ost["robust_chi2"])) / np.array(ost["robust_chi2"])
    assert rel.max() < 1e-8 and rrel.max() < 1e-8, (rel.max(), rrel.max())
    assert np.abs(P - oP).max() < 1e-7


def test_multifrontal_and_multigrid_paths_agree(monkeypatch):
    g = chain_graph(3000, 1200, seed=11)
    d1, done1, st1, P1 = run(g.arrays())
    monkeypatch.setenv("SGO_MFRONT", "0")
    d2, done2, st2, P2 = run(g.arrays(), pcg_tol=1e-11)
    assert d1.startswith("multifrontal_cholesky") and d2.startswith("pcg_amg") and "SGO_MFRONT" not in d2
    assert done1 == done2 == 20 and max(st2["pcg_iters"]) > 0 and st1["pcg_iters"] == [0] * 20
    rel = np.abs(np.array(st1["chi2"]) / np.array(st2["chi2"]) - 1)
    assert rel.max() < 1e-6 and rel[-5:].max() < 1e-10       # (the early iterates fall by orders of magnitude per step: PCG's tolerance shows)
    assert np.abs(P1 - P2).max() < 1e-6


def test_two_runs_are_bitwise_identical():
    g = chain_graph(2500, 700, seed=12)
    a = run(g.arrays())
    b = run(g.arrays())
    assert a[2]["chi2"] == b[2]["chi2"] and np.array_equal(a[3], b[3])


def test_duplicate_edges_fixed_poses_a_hub_and_edges_between_fixed_poses():
    """Several edges on one pair (summed in edge order), a pose with many incident closures, fixed poses in the middle
    of the chain (their edges contribute to the free endpoint's diagonal block and right-hand side only), edges between
    two fixed poses (chi2 only)."""
    g = chain_graph(600, 150, seed=13)
    ei, ej, meas, info, phi = [a.copy() for a in (g.ei, g.ej, g.meas, g.info, g.phi)]
    clo = np.arange(599, g.E)
    hub = int(ei[clo[0]])
    others = np.array([50, 120, 200, 201, 310, 388, 455, 590], dtype=np.int32)
    others = others[others != hub]

    def rel_meas(a, b):
        d = g.truth[b] - g.truth[a]
        c, s = np.cos(g.truth[a, 2]), np.sin(g.truth[a, 2])
        return [c * d[0] + s * d[1], -s * d[0] + c * d[1], d[2]]

    extra_i = [hub] * len(others) + [hub, hub]
    extra_j = list(others) + [int(others[0]), int(others[0])]
    em = np.array([rel_meas(a, b) for a, b in zip(extra_i, extra_j)])
    em[len(others)] += 0.01          # the tripled pair carries slightly different measurements
    em[len(others) + 1] -= 0.02
    ei = np.concatenate([ei, np.array(extra_i, np.int32)])
    ej = np.concatenate([ej, np.array(extra_j, np.int32)])
    meas = np.concatenate([meas, em])
    info = np.concatenate([info, np.tile(g.info[clo[0]], (len(extra_i), 1))])
    phi = np.concatenate([phi, np.full(len(extra_i), 10.0)])
    fixed = g.fixed.copy()
    fixed[[0, 1, 250, 251]] = True            # edges (0,1) and (250,251) join two fixed poses
    args = [g.poses, fixed, ei, ej, meas, info, phi]
    desc, done, st, P = run(args, iters=10)
    assert desc.startswith("multifrontal_cholesky"), desc
    oP, ost = _oracle().gauss_newton(*args, iters=10)
    assert done == ost["iters_done"] == 10
    for k in range(11):
        assert abs(st["chi2"][k] - ost["chi2"][k]) <= 1e-8 * ost["chi2"][k], k
    assert np.abs(P - oP).max() < 1e-7
    assert np.array_equal(P[[0, 1, 250, 251]], g.poses[[0, 1, 250, 251]])


def test_two_components_with_a_fixed_pose_each():
    """Halves that no edge connects: a front without pivots merges them (or the tree has two roots' worth of work under one)."""
    g = chain_graph(400, 60, seed=14)
    keep = ~(((g.ei < 200) & (g.ej >= 200)) | ((g.ej < 200) & (g.ei >= 200)))
    fixed = g.fixed.copy()
    fixed[200] = True
    args = [g.poses, fixed, g.ei[keep], g.ej[keep], g.meas[keep], g.info[keep], g.phi[keep]]
    desc, done, st, P = run(args, iters=8)
    assert desc.startswith("multifrontal_cholesky"), desc
    oP, ost = _oracle().gauss_newton(*args, iters=8)
    assert done == ost["iters_done"] == 8
    for k in range(9):
        assert abs(st["chi2"][k] - ost["chi2"][k]) <= 1e-8 * ost["chi2"][k], k
    assert np.abs(P - oP).max() < 1e-7


def test_graphs_that_do_not_qualify_keep_the_multigrid_path_and_say_why(monkeypatch):
    g = synth.config("C2")                       # four edges per pose
    with capi.Optimizer(0) as o:
        o.set_graph(*g.arrays())
        d = o.solver_description()
        assert d.startswith("pcg_amg") and "multifrontal path not used: 4.00 edges per free pose" in d, d
    g = chain_graph(900, 200, seed=15)
    monkeypatch.setenv("SGO_MFRONT_CRIT_MFLOP", "0.01")      # a budget no tree meets
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(*g.arrays())
        d = o.solver_description()
        assert d.startswith("pcg_amg") and "Mflop on the critical path" in d, d
        done, st = o.optimize(3)
        assert done == 3 and max(st["pcg_iters"]) > 0
    monkeypatch.delenv("SGO_MFRONT_CRIT_MFLOP")
    monkeypatch.setenv("SGO_MFRONT_ROWS", "500")
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(*g.arrays())
        assert "multifrontal path not used: 899 free poses > 500" in o.solver_description()
    monkeypatch.delenv("SGO_MFRONT_ROWS")
    with capi.Optimizer(0, solver=capi.SOLVER_PCG_BJ) as o:       # an explicit PCG solver is honoured
        o.set_graph(*g.arrays())
        assert o.solver_description().startswith("pcg_block_jacobi")


def test_indefinite_hessian_fails_like_g2o_and_keeps_the_estimates():
    """LinearSolverEigen::solve returning false: optimize() returns 0, the step is not applied."""
    g = chain_graph(700, 150, seed=16)
    info = g.info.copy()
    info[:, [0, 3, 5]] *= -1.0
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(g.poses, g.fixed, g.ei, g.ej, g.meas, info, g.phi)
        assert o.solver_description().startswith("multifrontal_cholesky")
        rc, st = o.optimize(5)
        assert rc == 0 and st["iters_done"] == 0
        assert "multifrontal factorisation failed" in o.last_error() and "not positive definite" in o.last_error()
        assert np.array_equal(o.get_poses(), g.poses)
        c, _ = o.chi2()
        assert abs(st["chi2"][0] - c) <= 1e-12 * abs(c)
        # the context stays usable: the same call again fails the same way, a good graph then works
        rc2, _ = o.optimize(2)
        assert rc2 == 0
        o.set_graph(*g.arrays())
        done, _ = o.optimize(2)
        assert done == 2


def test_disconnected_free_component_is_singular_and_fails_cleanly():
    """Two chains, only one of them tied to the fixed pose: the other's block is singular (gauge freedom)."""
    g = chain_graph(300, 0, seed=17)
    keep = ~((g.ei == 149) & (g.ej == 150)) & ~((g.ei == 150) & (g.ej == 149))
    args = [g.poses, g.fixed, g.ei[keep], g.ej[keep], g.meas[keep], g.info[keep], g.phi[keep]]
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(*args)
        assert o.solver_description().startswith("multifrontal_cholesky")
        rc, st = o.optimize(3)
        P = o.get_poses()
    # the singular block's last pivot is zero up to rounding: the positive-definiteness test catches it in this or a later
    # iteration (the estimates then stay at the last applied update); non-finite poses are never written
    assert np.isfinite(P).all()
    assert rc == 0 and st["iters_done"] < 3


def test_zero_iterations_continuation_and_set_poses():
    g = chain_graph(1200, 300, seed=18)
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(*g.arrays())
        done, st = o.optimize(0)
        c, r = o.chi2()
        assert done == 0 and st["chi2"][0] == pytest.approx(c, rel=1e-13) and st["robust_chi2"][0] == pytest.approx(r, rel=1e-13)
        d1, s1 = o.optimize(3)
        d2, s2 = o.optimize(4)                   # continues from the poses the first call left
        Pa = o.get_poses()
        assert s2["chi2"][0] == s1["chi2"][3]
        o.set_poses(g.poses)
        d3, s3 = o.optimize(7)
        assert d1 == 3 and d2 == 4 and d3 == 7
        assert s3["chi2"][:4] == s1["chi2"] and s3["chi2"][3:] == s2["chi2"]
        assert np.array_equal(o.get_poses(), Pa)


def test_single_step_entry_points_still_work_on_a_multifrontal_graph():
    """sgo_linearize / sgo_solve build the multigrid hierarchy on demand; their solution is the factorisation's step."""
    g = chain_graph(900, 200, seed=19)
    with capi.Optimizer(0, direct_rows=1, pcg_tol=1e-11) as o:
        o.set_graph(*g.arrays())
        assert o.solver_description().startswith("multifrontal_cholesky")
        b, _, c0, _ = o.linearize()
        x, it, relres = o.solve()
        assert it > 0 and relres <= 1e-10
        assert np.linalg.norm(b - o.hessian_apply(x)) <= 1e-8 * np.linalg.norm(b)
        done, st = o.optimize(1)
        P = o.get_poses()
    assert done == 1 and st["chi2"][0] == pytest.approx(c0, rel=1e-12)
    free = ~g.fixed
    step = P[free] - g.poses[free]
    step[:, 2] = (step[:, 2] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(step - x).max() <= 1e-6 * max(1.0, np.abs(x).max())


def test_growth_through_the_update_entry_point_is_a_full_set_up_and_says_so():
    """sgo_update_graph_se2 on a graph of this path: the analysis is cheap, the call is sgo_set_graph_se2."""
    base, steps, g = synth.append_session(1500, 2200, 1, 10, 23, info_mode="full", phi=0.75)
    V1 = steps[0]["V"]
    fixed = np.zeros(V1, dtype=bool)
    fixed[:base.V] = base.fixed
    cat = lambda k: np.concatenate([getattr(base, k), steps[0][k]])   # noqa: E731
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(*base.arrays())
        assert o.solver_description().startswith("multifrontal_cholesky")
        o.optimize(5)
        P = np.zeros((V1, 3))
        P[:base.V] = o.get_poses()
        synth.chain_init(P, g.meas[: g.V - 1], base.V, V1 - 1)
        o.update_graph(P, fixed, cat("ei"), cat("ej"), cat("meas"), cat("info"), cat("phi"), base.E)
        d = o.solver_description()
        assert d.startswith("multifrontal_cholesky") and "takes the multifrontal path" in d, d
        done, st = o.optimize(5)
        Pu = o.get_poses()
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(P, fixed, cat("ei"), cat("ej"), cat("meas"), cat("info"), cat("phi"))
        done2, st2 = o.optimize(5)
        assert done == done2 == 5 and st["chi2"] == st2["chi2"] and np.array_equal(Pu, o.get_poses())


def test_update_after_a_single_step_entry_point_is_still_a_full_set_up():
    """sgo_linearize on a multifrontal graph builds the PCG structures on demand and clears the pending flags; a
    following sgo_update_graph_se2 must NOT become an overlay (sgo_optimize_gn keeps taking the factorisation path, which would
    leave the appended poses and edges out): it is a full set-up, the description keeps naming the factorisation, and the
    iterates equal those of a fresh sgo_set_graph_se2 bit for bit."""
    base, steps, g = synth.append_session(1500, 2200, 1, 10, 29, info_mode="full", phi=0.75)
    V1 = steps[0]["V"]
    fixed = np.zeros(V1, dtype=bool)
    fixed[:base.V] = base.fixed
    cat = lambda k: np.concatenate([getattr(base, k), steps[0][k]])   # noqa: E731
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(*base.arrays())
        assert o.solver_description().startswith("multifrontal_cholesky")
        o.linearize()   # ensure_amg: hierarchy on demand
        d = o.solver_description()
        assert d.startswith("multifrontal_cholesky") and "single-step entry points: pcg_" in d, d
        P = np.zeros((V1, 3))
        P[:base.V] = o.get_poses()
        synth.chain_init(P, g.meas[: g.V - 1], base.V, V1 - 1)
        o.update_graph(P, fixed, cat("ei"), cat("ej"), cat("meas"), cat("info"), cat("phi"), base.E)
        d = o.solver_description()
        assert d.startswith("multifrontal_cholesky") and "incremental overlay" not in d and "full set-up" in d, d
        assert o.n_free == V1 - int(fixed.sum())
        done, st = o.optimize(5)
        Pu = o.get_poses()
        c_after, _ = o.chi2()
    with capi.Optimizer(0, direct_rows=1) as o:
        o.set_graph(P, fixed, cat("ei"), cat("ej"), cat("meas"), cat("info"), cat("phi"))
        done2, st2 = o.optimize(5)
        assert done == done2 == 5 and st["chi2"] == st2["chi2"] and np.array_equal(Pu, o.get_poses())
        assert st["chi2"][-1] == pytest.approx(c_after, rel=1e-12)   # the history covers the appended edges too
    # the appended poses moved (they were optimised)
    assert np.abs(Pu[base.V:] - P[base.V:]).max() > 0.0


def test_profile_names_the_path_and_stats_carry_device_times():
    g = chain_graph(1500, 400, seed=20)
    with capi.Optimizer(0, direct_rows=1, profile=1) as o:
        o.set_graph(*g.arrays())
        o.profile_reset()
        done, st = o.optimize(20)
        prof = o.kernel_profile()
    name = [k for k in prof if k.startswith("k_mf_edges")]
    assert done == 20 and len(name) == 1 and prof[name[0]]["launches"] == 1 and prof[name[0]]["ms"] > 0
    assert all(0 < s < 1e-2 for s in st["seconds"]) and all(0 < a < b for a, b in zip(st["seconds_linearize"], st["seconds"]))
    assert abs(sum(st["seconds"]) * 1e3 - prof[name[0]]["ms"]) < 0.5 * prof[name[0]]["ms"]
