"""The multigrid set-up ON THE DEVICE (sparse_gslam_amd/csrc/sgo_amg_dev.inc, round 6): what a rebuild of the hierarchy inside
sgo_optimize_gn runs instead of the host's aggregation and pattern work (sgo_amg_host.cpp) -- g2o's counterpart is the symbolic
analysis LinearSolverEigen redoes per optimize() (graphs.cpp:19, slc.cpp:286-287).

1. Patterns (the default device set-up: the host's greedy aggregation, everything else on the device): the device-made hierarchy must
   be the host-made one bit for bit -- same sizes of P, A P, P^T A P and of every product list (the solver description prints them),
   the same PCG iteration counts, bitwise the same chi2 history -- on every kind of level: smoothed, smoothed with the filtered
   operator, tentative (K-cycle), folded level 0; with aggregates made anew and with the aggregates of the replaced hierarchy kept.
2. Aggregation on the device as well (SGO_AMG_AGG=device, opt-in: a distance-2 independent set by hashed priorities packs less regularly
   than the host's greedy walk along the trajectory and its hierarchies need 40-50 % more PCG iterations -- measured, NOTES.md section
   28 --): what is checked is what the solver needs from it -- every solve converges, the iterates are the direct-solver goldens' to
   1e-6, the PCG counts stay within a factor of two of the host hierarchy's.
3. The rebuilds inside a call from BASELINE.md's literal dead-reckoned start: the same iterates and counts as with the host's
   rebuilds, in less time."""
import os

import numpy as np
import pytest

from sparse_gslam_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(g, iters, env, monkeypatch, **opts):
    for k in ("SGO_AMG_FORCE_REBUILD", "SGO_AMG_KEEP_AGG", "SGO_AMG_SETUP"):
        monkeypatch.delenv(k, raising=False)
    with capi.Optimizer(0, direct_rows=0, **opts) as opt:
        opt.set_graph(*g.arrays())          # (the host set-up, as always at sgo_set_graph_se2)
        d0 = opt.solver_description()
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        done, st = opt.optimize(iters)
        desc = opt.solver_description()
    for k in env:
        monkeypatch.delenv(k, raising=False)
    return done, st, d0, desc


CASES = {
    "C2": lambda: synth.config("C2"),                                                        # smoothed levels, level 0 folded
    "C2_odom": lambda: synth.config("C2", init="odom"),                                      # filtered smoothing on level 0
    "30k_300k": lambda: synth.manhattan(30000, 300000, seed=3, info_mode="full"),           # smoothed, level 0 not folded
    "20k_random": lambda: synth.manhattan(20000, 100000, seed=5, p_random=0.05),            # tentative levels, K-cycle
    "chain": lambda: synth.manhattan(8000, 8400, seed=6, info_mode="full"),                 # few closures (multifrontal off below)
}


@pytest.mark.parametrize("keep", [False, True])
@pytest.mark.parametrize("name", list(CASES))
def test_device_patterns_from_the_hosts_aggregates_are_the_hosts_hierarchy(name, keep, monkeypatch):
    monkeypatch.setenv("SGO_MFRONT", "0")
    g = CASES[name]()
    base = {"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_KEEP_AGG": "1" if keep else "0"}
    dh, sh, d0h, desch = _run(g, 4, dict(base, SGO_AMG_SETUP="host"), monkeypatch)
    dd, sd, d0d, descd = _run(g, 4, dict(base, SGO_AMG_SETUP="device"), monkeypatch)
    assert dh == 4 and dd == 4
    strip = lambda d: d.split("; last sgo_optimize_gn")[0]
    assert d0h == d0d
    assert strip(desch) == strip(descd), (desch, descd)
    assert strip(desch).split("; direct path")[0] == strip(d0h).split("; direct path")[0]   # (kept aggregates, same values: the same hierarchy)
    assert sh["pcg_iters"][:4] == sd["pcg_iters"][:4], (sh["pcg_iters"], sd["pcg_iters"])
    assert list(sh["chi2"][:5]) == list(sd["chi2"][:5])                                       # bitwise
    assert list(sh["robust_chi2"][:5]) == list(sd["robust_chi2"][:5])


@pytest.mark.parametrize("name", ["C2", "C2_odom", "C4", "30k_300k", "20k_random"])
def test_set_graph_with_device_patterns_is_the_host_set_up(name, monkeypatch):
    """sgo_set_graph_se2 itself (the default since round 6: level 0's aggregation on the set-up pipeline's helper thread where there
    is one, every pattern and list on the device) against SGO_AMG_SETUP=host: the same hierarchy, bitwise the same optimize(20)."""
    monkeypatch.setenv("SGO_MFRONT", "0")
    g = synth.config("C4") if name == "C4" else CASES[name]()
    out = {}
    for mode in ("host", "device"):
        monkeypatch.setenv("SGO_AMG_SETUP", mode)
        with capi.Optimizer(0, direct_rows=0) as opt:
            opt.set_graph(*g.arrays())
            desc = opt.solver_description()
            done, st = opt.optimize(20)
            out[mode] = (done, st, desc)
    monkeypatch.delenv("SGO_AMG_SETUP")
    assert out["host"][0] == 20 and out["device"][0] == 20
    assert out["host"][2] == out["device"][2], (out["host"][2], out["device"][2])
    assert out["host"][1]["pcg_iters"][:20] == out["device"][1]["pcg_iters"][:20]
    assert list(out["host"][1]["chi2"][:21]) == list(out["device"][1]["chi2"][:21])
    assert list(out["host"][1]["robust_chi2"][:21]) == list(out["device"][1]["robust_chi2"][:21])


@pytest.mark.parametrize("name", ["C2", "C4"])
def test_device_aggregation_hierarchy_solves_to_the_goldens(name, monkeypatch):
    monkeypatch.setenv("SGO_MFRONT", "0")
    f = np.load(os.path.join(GOLDEN, f"{name}_direct.npz"))
    g = synth.config(name)
    dh, sh, _, desch = _run(g, 20, {"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_SETUP": "host"}, monkeypatch)
    dd, sd, _, descd = _run(g, 20, {"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_SETUP": "device", "SGO_AMG_AGG": "device"}, monkeypatch)
    monkeypatch.delenv("SGO_AMG_AGG", raising=False)
    assert dh == 20 and dd == 20 and all(sd["pcg_converged"][:20])
    rel = max(abs(sd["chi2"][k] - f["chi2"][k]) / f["chi2"][k] for k in range(21))
    assert rel <= 1e-6, rel
    print(name, "host", desch.split("; direct")[0], sh["pcg_iters"])
    print(name, "device", descd.split("; direct")[0], sd["pcg_iters"])
    assert sum(sd["pcg_iters"][:20]) <= 2.0 * sum(sh["pcg_iters"][:20]), (sd["pcg_iters"], sh["pcg_iters"])


def test_device_rebuilds_from_the_dead_reckoned_start(monkeypatch):
    """BASELINE.md's literal start on C4: the call rebuilds its hierarchy every few iterations.  With the rebuilds' patterns made on the
    device the call is the SAME call -- counts and chi2 history bit for bit -- and shorter."""
    g = synth.config("C4", init="odom")
    res = {}
    for mode in ("host", "rebuilds", "host", "rebuilds"):
        monkeypatch.setenv("SGO_AMG_SETUP", mode)
        with capi.Optimizer(0) as opt:
            opt.set_graph(*g.arrays())
            done, st = opt.optimize(20)
            res[mode] = (done, st, opt.solver_description())
    monkeypatch.delenv("SGO_AMG_SETUP")
    for mode, (done, st, desc) in res.items():
        assert done == 20 and all(st["pcg_converged"][:20]), (mode, st["pcg_iters"])
        print(mode, "call ms", 1e3 * st["seconds_total"], "pcg", st["pcg_iters"])
    assert res["rebuilds"][1]["pcg_iters"][:20] == res["host"][1]["pcg_iters"][:20]
    assert list(res["rebuilds"][1]["chi2"][:21]) == list(res["host"][1]["chi2"][:21])
    assert res["rebuilds"][1]["seconds_total"] <= 0.9 * res["host"][1]["seconds_total"]


def test_a_device_set_up_that_fails_falls_back_to_the_host_set_up(monkeypatch, capfd):
    """Out of device memory for the sort buffers, say (here: a test hook): the host set-up, which needs none, takes over -- at
    sgo_set_graph_se2 (where the helper thread has made level 0's aggregation only, which the host set-up cannot use) and for a
    rebuild --, and the call is the host call."""
    monkeypatch.setenv("SGO_MFRONT", "0")
    g = synth.config("C4")
    out = {}
    for mode, fail in (("host", "0"), ("device", "1")):
        monkeypatch.setenv("SGO_AMG_SETUP", mode)
        monkeypatch.setenv("SGO_TEST_FAIL_DEVICE_SETUP", fail)
        monkeypatch.setenv("SGO_AMG_FORCE_REBUILD", "1")
        monkeypatch.setenv("SGO_VERBOSE", "1")
        with capi.Optimizer(0, direct_rows=0) as opt:
            opt.set_graph(*g.arrays())
            desc = opt.solver_description()
            done, st = opt.optimize(3)
            out[mode] = (done, st, desc, capfd.readouterr().err)
    for k in ("SGO_AMG_SETUP", "SGO_TEST_FAIL_DEVICE_SETUP", "SGO_AMG_FORCE_REBUILD", "SGO_VERBOSE"):
        monkeypatch.delenv(k)
    assert out["device"][3].count("device set-up failed (test hook): host set-up") == 2, out["device"][3][-2000:]
    assert out["device"][2].startswith("pcg_amg") and out["device"][2] == out["host"][2]
    assert out["device"][0] == 3 and out["host"][0] == 3
    assert out["host"][1]["pcg_iters"][:3] == out["device"][1]["pcg_iters"][:3]
    assert list(out["host"][1]["chi2"][:4]) == list(out["device"][1]["chi2"][:4])
