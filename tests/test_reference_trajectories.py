"""C1-sized pose graphs on the keyframe trajectories the reference ships (intel-lab, aces; fixture
tests/golden/ref_trajectories.npz made by scripts/make_traj_fixture.py from
src/sparse_gslam/datasets/*/ *30pts.txt): real revisit topology instead of a random walk."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no
from sparse_gslam_amd import capi, synth


def test_fixture_is_the_shipped_trajectory():
    intel, aces = synth.reference_trajectory("intel_lab"), synth.reference_trajectory("aces")
    assert intel.shape == (1051, 3) and aces.shape == (440, 3)
    # keyframes are ~0.5 m / 30 degrees apart (drone.cpp:112): steps stay below 1.5 m
    for P in (intel, aces):
        step = np.hypot(*np.diff(P[:, :2], axis=0).T)
        assert step.max() < 1.5 and np.median(step) > 0.2
    g = synth.config("C1i")
    assert (g.V, g.E) == (1051, 1110) and g.fixed[0] and not g.fixed[1:].any()
    assert (g.phi[:1050] < 0).all() and (g.phi[1050:] == 10.0).all()
    sep = np.abs(g.ei[1050:].astype(int) - g.ej[1050:].astype(int))
    assert sep.min() > 40                       # closures are revisits, not neighbours


@pytest.mark.parametrize("name", ["C1i", "C1a"])
def test_two_oracles_agree_on_reference_trajectories(name):
    g = synth.config(name)
    Pn, sn = no.gauss_newton(*g.arrays(), iters=10)
    Pc, sc = co.gauss_newton(*g.arrays(), iters=10)
    assert sc["iters_done"] == 10
    rel = np.abs(np.array(sn["chi2"]) - np.array(sc["chi2"])) / np.array(sc["chi2"])
    assert rel.max() < 1e-9
    assert np.abs(Pn - Pc).max() < 1e-7
    assert sc["chi2"][-1] < sc["chi2"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("name,init", [("C1i", "incremental"), ("C1a", "incremental"), ("C1a", "odom")])
def test_gpu_matches_oracle_on_reference_trajectories(name, init):
    g = synth.config(name, init=init)
    with capi.Optimizer(0) as opt:
        opt.set_graph(*g.arrays())
        done, st = opt.optimize(20)
        P = opt.get_poses()
    oP, ost = co.gauss_newton(*g.arrays(), iters=20)
    assert done == ost["iters_done"] == 20
    rel = np.abs(np.array(st["chi2"]) - np.array(ost["chi2"])) / np.array(ost["chi2"])
    assert rel.max() < 1e-6, rel                 # every iterate, BASELINE.json's bound
    assert np.abs(P - oP).max() < 1e-5
