"""Randomised parity sweep through ONE context (set_graph / optimize / read-back repeatedly): graph
size, density, share of long-range edges, information shape, initialisation and extra fixed vertices
drawn at random; every iterate's chi2 against the CPU direct-solver oracle.  (scripts/stress.py is the
long version with graphs up to 20 000 poses.)"""
import numpy as np
import pytest

from sparse_gslam_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_random_graphs_match_oracle_through_one_context():
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    shapes = [(60, 6.0), (300, 1.05), (450, 3.0), (1200, 1.5), (3000, 1.05), (3000, 6.0), (800, 1.0),
              (2000, 3.0), (1500, 1.2), (401, 2.0)]
    with capi.Optimizer(0) as opt:
        for case, (V, dens) in enumerate(shapes):
            E = max(V - 1 + 2, int(dens * V))
            g = synth.manhattan(V=V, E=E, seed=int(rng.integers(1, 10**6)), p_random=float(rng.choice([0.0, 0.05, 0.3])),
                                info_mode=str(rng.choice(["diag", "full"])), phi=float(rng.choice([1.0, 10.0])),
                                init=str(rng.choice(["incremental", "odom"])) if V <= 1200 else "incremental")
            if case % 3 == 0:
                g.fixed[rng.integers(1, V, 3)] = True
            opt.set_graph(*g.arrays())
            done, st = opt.optimize(6)
            P = opt.get_poses()
            oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=6)
            assert done == ost["iters_done"] == 6, (case, V, E)
            floor = 1e-9 * ost["chi2"][0]
            rel = max(abs(a - b) / max(b, floor) for a, b in zip(st["chi2"], ost["chi2"]))
            # 1e-6 relative is BASELINE.json's bound and holds for every iterate, also on the chain-like
            # graphs of thousands of poses (kappa(H) beyond 1e10), which run at a tightened PCG tolerance
            assert rel <= 1e-6, (case, V, E, rel)
            assert np.isfinite(P).all() and np.abs(P - oP).max() < 1e-4, (case, V, E)
