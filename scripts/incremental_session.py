#!/usr/bin/env python3
"""Replay of the reference's back-end usage pattern (src/sparse_gslam/src/submap_loop_closer.cpp:205-288):
the pose graph grows along the trajectory, and after every accepted loop closure the WHOLE graph is
re-initialised and optimised with optimize(20).  Reports the back-end optimisation latency per
closure for libsgo (set_graph + optimize + get_poses, i.e. what SparseOptimizer::initializeOptimization
+ optimize cost through the compat header) and for the single-thread CPU oracle (direct LDL^T).

    python scripts/incremental_session.py V n_closures [seed]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import c_oracle  # noqa: E402
from sparse_gslam_amd import capi, synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = synth.manhattan(V, V - 1 + NC, seed=seed, info_mode="full", init="odom", phi=10.0)
odo = np.arange(V - 1)
clo = np.arange(V - 1, g.E)
clo = clo[np.argsort(np.maximum(g.ei[clo], g.ej[clo]))]       # closures in the order they would be found
poses_gpu = g.poses.copy()
poses_cpu = g.poses.copy()
t_gpu, t_cpu, sizes = [], [], []
with capi.Optimizer(0) as opt:
    for k, c in enumerate(clo):
        last = int(max(g.ei[c], g.ej[c]))                        # trajectory has reached this pose
        edges = np.concatenate([odo[:last], clo[: k + 1]])
        edges = edges[(g.ei[edges] <= last) & (g.ej[edges] <= last)]
        sl = slice(0, last + 1)
        args = lambda P: (P[sl], g.fixed[sl], g.ei[edges], g.ej[edges], g.meas[edges], g.info[edges], g.phi[edges])  # noqa: E731
        t = time.perf_counter()
        opt.set_graph(*args(poses_gpu))
        done, st = opt.optimize(20)
        poses_gpu[sl] = opt.get_poses()
        t_gpu.append(time.perf_counter() - t)
        t = time.perf_counter()
        P, ost = c_oracle.gauss_newton(*args(poses_cpu), iters=20)
        poses_cpu[sl] = P
        t_cpu.append(time.perf_counter() - t)
        sizes.append((last + 1, len(edges)))
        rel = abs(st["chi2"][-1] - ost["chi2"][-1]) / max(ost["chi2"][-1], 1e-30)
        assert done == 20 and (rel < 1e-5 or ost["chi2"][-1] < 1e-9), (k, rel)
tg, tc = 1e3 * np.array(t_gpu), 1e3 * np.array(t_cpu)
print(f"V={V} closures={NC}: graph grows to {sizes[-1]}; per-closure optimize(20) latency [ms]")
print(f"  libsgo (set_graph+optimize+get_poses): mean {tg.mean():.2f}  median {np.median(tg):.2f}  max {tg.max():.2f}")
print(f"  CPU oracle (analysis + 20 x LDL^T)   : mean {tc.mean():.2f}  median {np.median(tc):.2f}  max {tc.max():.2f}")
print(f"  final poses max |gpu - cpu| = {np.abs(poses_gpu - poses_cpu).max():.2e}")
