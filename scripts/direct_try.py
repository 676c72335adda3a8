import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
from oracle import c_oracle
for (V, NC) in [(50, 3), (300, 10), (1000, 30), (2000, 45)]:
    g = synth.manhattan(V, V - 1 + NC, seed=1, info_mode="full", init="odom", phi=10.0)
    with capi.Optimizer(0) as o:
        o.set_graph(*g.arrays())
        print(V, NC, o.solver_description())
        for rep in range(3):
            o.set_poses(g.poses)
            t = time.perf_counter(); done, st = o.optimize(20); dt = time.perf_counter() - t
        P = o.get_poses()
        t = time.perf_counter(); o.set_graph(*g.arrays()); tset = time.perf_counter() - t
    Pc, ost = c_oracle.gauss_newton(*g.arrays(), iters=20)
    rel = max(abs(st["chi2"][k] - ost["chi2"][k]) / max(ost["chi2"][k], 1e-30) for k in range(21))
    print(f"  done={done} optimize(20) {1e3*dt:.2f} ms (device {1e3*sum(st['seconds']):.2f}, lin {1e3*sum(st['seconds_linearize']):.2f}) set_graph {1e3*tset:.2f} ms; chi2 {st['chi2'][0]:.6e}->{st['chi2'][-1]:.6e} max rel chi2 err {rel:.2e} pose err {np.abs(P-Pc).max():.2e}")
