import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
g = synth.manhattan(100000, 1000000, seed=9, info_mode="full")
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    done, st = o.optimize(20)
    print("done", done, "pcg", st["pcg_iters"], "gn ms", [round(1e3*s,2) for s in st["seconds"]][:20])
