import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
cands = []
for seed in range(1, 9):
    cands.append(dict(V=3000, E=3150, seed=seed, p_random=0.3, info_mode="full", phi=10.0))
    cands.append(dict(V=1200, E=1800, seed=seed, p_random=0.05, info_mode="full", phi=1.0, init="odom"))
    cands.append(dict(V=3000, E=4500, seed=seed, p_random=0.3, info_mode="diag", phi=1.0, init="odom"))
for kw in cands:
    g = synth.manhattan(**kw)
    with capi.Optimizer(0, direct_rows=0, verbose=1) as o:
        o.set_graph(*g.arrays())
        done, st = o.optimize(20)
    print("CASE", kw, "done", done, "pcg", st["pcg_iters"], flush=True)
