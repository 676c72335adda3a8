"""g2o text format and CARMEN .result round trips."""
import numpy as np

from sparse_gslam_amd import graph_io, synth


def test_g2o_round_trip(tmp_path):
    g = synth.manhattan(60, 110, seed=3, info_mode="full")
    p = tmp_path / "g.g2o"
    graph_io.write_g2o(str(p), g)
    h = graph_io.read_g2o(str(p), loop_phi=1.0)
    assert h.V == g.V and h.E == g.E
    for a, b in zip(g.arrays(), h.arrays()):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_g2o_reader_compacts_ids_and_fixes_first(tmp_path):
    p = tmp_path / "s.g2o"
    p.write_text("VERTEX_SE2 10 0 0 0\nVERTEX_SE2 30 2 0 0.1\nVERTEX_SE2 20 1 0 0\n"
                 "EDGE_SE2 10 20 1 0 0 100 0 0 100 0 400\nEDGE_SE2 20 30 1 0 0.1 100 0 0 100 0 400\n"
                 "EDGE_SE2 10 30 2 0 0.1 50 1 2 60 3 200\n")
    g = graph_io.read_g2o(str(p), loop_phi=0.75)
    assert g.V == 3 and list(g.meta["ids"]) == [10, 20, 30]
    assert np.array_equal(g.poses[:, 0], [0, 1, 2]) and g.fixed.tolist() == [True, False, False]
    assert g.ei.tolist() == [0, 1, 0] and g.ej.tolist() == [1, 2, 2]
    assert g.phi.tolist() == [-1.0, -1.0, 0.75]
    assert g.info[2].tolist() == [50, 1, 2, 60, 3, 200]


def test_carmen_result_round_trip(tmp_path):
    P = np.array([[0, 0, 0], [1.5, -2.25, 0.5], [3, 4, -3.0]])
    T = np.array([0.0, 0.5, 1.25])
    p = tmp_path / "t.result"
    graph_io.write_carmen_result(str(p), P, T)
    first = p.read_text().splitlines()[1]
    assert first == "FLASER 0 1.5 -2.25 0.5 1.5 -2.25 0.5 0.5 myhost 0.5"
    Q, S = graph_io.read_carmen_result(str(p))
    assert np.allclose(Q, P) and np.allclose(S, T)


# ---------------------------------------------------------------- the C++ side (compat header)
import os  # noqa: E402
import subprocess  # noqa: E402

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_cpp_load_save_round_trip_is_exact(tmp_path):
    """SparseOptimizer::load / save and EdgeSE2::read / write of the compat header (the reference's own
    read / write members are stubs, src/sparse_gslam/src/g2o_bindings/edge_se2_rhotheta.cpp:18-23): a file
    written by the Python writer is loaded and saved by the C++ side and read back value for value, ids and
    FIX records included; unknown record types are skipped.  No GPU involved."""
    g = synth.manhattan(80, 150, seed=5, info_mode="full")
    g.fixed[7] = True
    a, b = tmp_path / "a.g2o", tmp_path / "b.g2o"
    graph_io.write_g2o(str(a), g)
    with open(a, "a") as f:
        f.write("VERTEX_XY 900 1.0 2.0\n# a comment\n")
    subprocess.check_call(["make", "-s", "-C", CPP, "graph_io"])
    out = subprocess.run([os.path.join(CPP, "graph_io"), str(a), str(b)], capture_output=True, text=True, check=True)
    V, E, nfixed = map(int, out.stdout.split())
    assert (V, E, nfixed) == (g.V, g.E, int(g.fixed.sum()))
    assert "skipped 1 records" in out.stderr
    h = graph_io.read_g2o(str(b), loop_phi=1.0, fix_first=False)
    for x, y in zip(g.arrays(), h.arrays()):
        assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.gpu
def test_cpp_loaded_graph_optimises_like_the_c_abi(tmp_path):
    """A .g2o file through the compat header's load + initializeOptimization + optimize(20) on the GPU equals
    the same graph through the C-ABI; the CARMEN result file has the reference's line format."""
    from sparse_gslam_amd import capi
    g = synth.manhattan(1500, 4000, seed=9, info_mode="full", phi=1.0)
    a, b, r = tmp_path / "a.g2o", tmp_path / "b.g2o", tmp_path / "t.result"
    graph_io.write_g2o(str(a), g)
    subprocess.check_call(["make", "-s", "-C", CPP, "graph_io"])
    out = subprocess.run([os.path.join(CPP, "graph_io"), str(a), str(b), "optimize", str(r)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    its, chi2, rchi2 = lines[1].split()
    gg = graph_io.read_g2o(str(a), loop_phi=1.0)
    with capi.Optimizer(0) as o:
        o.set_graph(*gg.arrays())
        done, st = o.optimize(20)
        P = o.get_poses()
    assert int(its) == done == 20
    assert abs(float(chi2) - st["chi2"][-1]) <= 1e-9 * st["chi2"][-1]
    assert abs(float(rchi2) - st["robust_chi2"][-1]) <= 1e-9 * st["robust_chi2"][-1]
    Q, T = graph_io.read_carmen_result(str(r))
    # the reference streams with the default precision: 6 significant digits per number
    assert Q.shape == P.shape and np.abs(Q - P).max() <= 1e-5 * max(1.0, np.abs(P).max()) * 10 and T[1] == 1.0
    saved = graph_io.read_g2o(str(b), loop_phi=1.0)
    assert np.abs(saved.poses - P).max() <= 1e-12
