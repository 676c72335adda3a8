import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
from oracle import c_oracle
for V in (4000, 8000, 12000, 16000):
    worst = []
    for seed in (1, 2, 3):
        for init in ("odom", "incremental"):
            g = synth.manhattan(V, V - 1 + 45, seed=seed, info_mode="full", init=init, phi=10.0)
            res = []
            for rows in (1 << 20, 0):
                with capi.Optimizer(0, direct_rows=rows) as o:
                    o.set_graph(*g.arrays())
                    done, st = o.optimize(20)
                    res.append((done, st["chi2"]))
            oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=20)
            rel = [max(abs(r[1][k] - ost["chi2"][k]) / ost["chi2"][k] for k in range(min(r[0], 20) + 1)) for r in res]
            worst.append((seed, init, res[0][0], res[1][0], rel[0], rel[1]))
    print(V, " | ".join(f"s{w[0]} {w[1][:4]} d{w[2]}/{w[3]} direct {w[4]:.1e} amg {w[5]:.1e}" for w in worst), flush=True)
