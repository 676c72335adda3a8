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
