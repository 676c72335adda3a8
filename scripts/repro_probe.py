import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from sparse_gslam_amd import capi, synth
os.environ["SGO_MFRONT"] = "0"
g = synth.config("C2")
def run(env):
    for k in ("SGO_AMG_FORCE_REBUILD", "SGO_AMG_KEEP_AGG", "SGO_AMG_SETUP"):
        os.environ.pop(k, None)
    with capi.Optimizer(0, direct_rows=0) as opt:
        opt.set_graph(*g.arrays())
        os.environ.update(env)
        d, st = opt.optimize(4)
    return st
a = run({"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_KEEP_AGG": "1", "SGO_AMG_SETUP": "host"})
b = run({"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_KEEP_AGG": "1", "SGO_AMG_SETUP": "host"})
c = run({})
d = run({})
e = run({"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_KEEP_AGG": "1", "SGO_AMG_SETUP": "device"})
f = run({"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_KEEP_AGG": "1", "SGO_AMG_SETUP": "device"})
for name, x in (("host-rebuild", a), ("host-rebuild again", b), ("no rebuild", c), ("no rebuild again", d), ("device", e), ("device again", f)):
    print(name.ljust(20), [repr(v) for v in x["chi2"][:5]], x["pcg_iters"][:4])
