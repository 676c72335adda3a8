import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
from oracle import c_oracle
for V, E in [(5489, 7629), (3500, 5600), (10000, 14000)]:
    g = synth.manhattan(V, E, seed=3, info_mode="full", init="incremental", phi=10.0)
    with capi.Optimizer(0) as o:
        ts = []
        for rep in range(3):
            t = time.perf_counter(); o.set_graph(*g.arrays()); t1 = time.perf_counter()
            done, st = o.optimize(20); t2 = time.perf_counter()
            ts.append((t1 - t, t2 - t1))
        d = o.solver_description()
    t = time.perf_counter(); oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=20); tc = time.perf_counter() - t
    rel = max(abs(st["chi2"][k] - ost["chi2"][k]) / ost["chi2"][k] for k in range(21))
    print(f"V={V} E={E}: {d[:70]}... set_graph {1e3*min(a for a,_ in ts):.1f} ms optimize(20) {1e3*min(b for _,b in ts):.1f} ms pcg {np.mean(st['pcg_iters']):.1f}; CPU oracle total {1e3*tc:.1f} ms; rel {rel:.1e}", flush=True)
