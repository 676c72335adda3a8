import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
import os
os.environ["SGO_AMG_LAZY"] = "0"
cases = {"C4": synth.config("C4"), "seed9": synth.manhattan(100000, 1000000, seed=9, info_mode="full"), "C2": synth.config("C2"),
         "C2full": synth.config("C2", info_mode="full"), "C4r": synth.config("C4r")}
for name, g in cases.items():
    with capi.Optimizer(0) as o:
        o.set_graph(*g.arrays())
        done, st = o.optimize(20)
    r = np.array(st["robust_chi2"]); c = np.array(st["chi2"])
    print(name, "pcg", st["pcg_iters"])
    print("   rel d robust", " ".join(f"{abs(r[k+1]-r[k])/r[k+1]:.1e}" for k in range(20)))
    print("   rel d chi2  ", " ".join(f"{abs(c[k+1]-c[k])/c[k+1]:.1e}" for k in range(20)))
