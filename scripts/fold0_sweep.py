#!/usr/bin/env python3
"""Where folding level 0 pays: ms per GN iteration with level 0 folded (SGO_AMG_FOLD0_ROWS large) and not (0) over graph sizes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from sparse_gslam_amd import capi, synth  # noqa: E402

for V, E in ((10000, 100000), (20000, 100000), (20000, 200000), (30000, 300000), (50000, 250000), (50000, 500000)):
    g = synth.manhattan(V, E, seed=7)
    out = []
    for rows in ("1000000", "0"):
        os.environ["SGO_AMG_FOLD0_ROWS"] = rows
        with capi.Optimizer(0) as o:
            o.set_graph(*g.arrays())
            o.optimize(20)
            o.set_poses(g.poses)
            done, st = o.optimize(20)
        out.append((float(np.median(st["seconds"][2:done])) * 1e3, sum(st["pcg_iters"][:done]) / done))
    print(f"V={V} E={E}: folded level 0 {out[0][0]:.3f} ms ({out[0][1]:.1f} its)  unfolded {out[1][0]:.3f} ms ({out[1][1]:.1f} its)", flush=True)
