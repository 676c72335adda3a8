"""sgo_set_graph_se2 on large graphs: the multigrid's host analysis of level 0 runs on a helper thread, fed by
strength weights computed straight from the edge list (k_row_strength), while the calling
thread lays out the level-0 storage (NOTES.md section 7).  The weights are the Frobenius norms of the same
Hessian blocks the serial set-up reads back from the assembled matrix, so both set-ups must build the same
hierarchy: same level sizes, same PCG iteration counts, same iterates to rounding.  Also covers fixed vertices
in the middle of the graph (slots without a block), duplicate edges, graph replacement while a helper result is
unused, and destruction of a context right after set_graph."""
import numpy as np
import pytest

from sparse_gslam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _graph(p_random):
    g = synth.manhattan(24000, 150000, seed=77, p_random=p_random, info_mode="full")
    g.fixed[[5000, 12345, 20001]] = True          # slots whose column is fixed: no block, no logical slot
    dup = np.arange(30000, 30040)                  # duplicate closures: two logical slots on one pair
    return (g.poses, g.fixed, np.concatenate([g.ei, g.ei[dup]]), np.concatenate([g.ej, g.ej[dup]]),
            np.concatenate([g.meas, g.meas[dup]]), np.concatenate([g.info, g.info[dup]]), np.concatenate([g.phi, g.phi[dup]]))


def _run(arrs, monkeypatch, pipeline):
    monkeypatch.setenv("SGO_SETUP_PIPELINE", "1" if pipeline else "0")
    with capi.Optimizer(0) as o:
        o.set_graph(*arrs)
        desc = o.solver_description()
        done, st = o.optimize(6)
        return desc, done, st, o.get_poses()


@pytest.mark.parametrize("p_random", [0.0, 0.05])   # smoothed hierarchy / tentative fallback
def test_helper_thread_setup_builds_the_same_hierarchy(monkeypatch, p_random):
    arrs = _graph(p_random)
    da, na, sa, Pa = _run(arrs, monkeypatch, True)
    db, nb, sb, Pb = _run(arrs, monkeypatch, False)
    assert na == nb == 6
    assert da == db, (da, db)                       # level sizes, block and product counts
    assert max(abs(x - y) for x, y in zip(sa["pcg_iters"], sb["pcg_iters"])) <= 1
    for x, y in zip(sa["chi2"], sb["chi2"]):
        assert abs(x - y) <= 1e-9 * y
    assert np.abs(Pa - Pb).max() <= 1e-7


def test_unused_helper_results_are_dropped_cleanly(monkeypatch):
    monkeypatch.setenv("SGO_SETUP_PIPELINE", "1")
    arrs = _graph(0.0)
    small = synth.manhattan(500, 1200, seed=3).arrays()
    o = capi.Optimizer(0)
    o.set_graph(*arrs)
    o.set_graph(*small)            # replaces the graph; nothing of the first set-up may linger
    done, _ = o.optimize(3)
    assert done == 3
    o.set_graph(*arrs)
    o.close()                      # destruction right after a set-up
    with capi.Optimizer(0) as o2:
        o2.set_graph(*arrs)
        done, st = o2.optimize(2)
        assert done == 2 and st["chi2"][-1] < st["chi2"][0]


@pytest.mark.parametrize("name", ["C2", "big"])
def test_device_made_product_lists_equal_the_host_lists(monkeypatch, name):
    """The lists of block products behind every entry of A P and P^T A P are made on the device from the host's
    patterns (k_ap_list / k_rap_list, prefix sums, wave groups); SGO_AMG_LISTS=host makes them on the host as before.
    Same products in the same order per target; the wave-group boundaries differ (the device packs chunks of 2048
    targets independently), and with them the association order of the wavefront segmented sums: identical
    hierarchies and product counts, iterates equal to rounding."""
    arrs = synth.config("C2", info_mode="full").arrays() if name == "C2" else _graph(0.0)
    res = []
    for mode in ("device", "host"):
        monkeypatch.setenv("SGO_AMG_LISTS", mode)
        with capi.Optimizer(0) as o:
            o.set_graph(*arrs)
            desc = o.solver_description()
            done, st = o.optimize(5)
            res.append((desc, done, st["chi2"], st["pcg_iters"], o.get_poses()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert res[0][1] == res[1][1] == 5
    assert res[0][3] == res[1][3]
    for x, y in zip(res[0][2], res[1][2]):
        assert abs(x - y) <= 1e-9 * y
    assert np.abs(res[0][4] - res[1][4]).max() <= 1e-8
