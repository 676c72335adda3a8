import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, os
os.environ["SGO_VERBOSE"]="1"
from sparse_gslam_amd import capi, synth
g=synth.config(sys.argv[1] if len(sys.argv)>1 else "C4")
with capi.Optimizer(0) as o:
    for r in range(3):
        t=time.perf_counter(); o.set_graph(*g.arrays()); print("set_graph %.1f ms"%(1e3*(time.perf_counter()-t)))
