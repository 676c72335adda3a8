#!/usr/bin/env python3
"""Calibration of the lagged refresh (sgo_solve.cpp): SGO_AMG_LAG_FORCE=N makes every solve after Gauss-Newton iteration N keep the
coarse operators of iteration N; per iteration, the movement measures the decision sees and the PCG count against the count with
a refresh before every solve.  SGO_VERBOSE=1 python scripts/lag_calib.py > out.txt 2>&1, then scripts/lag_calib_table.py out.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

os.environ["SGO_MFRONT"] = "0"
SHAPES = [
    ("C4", synth.config("C4")),
    ("C2", synth.config("C2")),
    ("20k / 60k", synth.manhattan(20000, 60000, seed=21)),
    ("50k / 250k", synth.manhattan(50000, 250000, seed=22)),
    ("50k / 250k full phi=10", synth.manhattan(50000, 250000, seed=22, info_mode="full", phi=10.0)),
    ("30k / 300k", synth.manhattan(30000, 300000, seed=23)),
]
for name, g in SHAPES:
    os.environ["SGO_AMG_LAG"] = "0"
    os.environ.pop("SGO_AMG_LAG_FORCE", None)
    with capi.Optimizer(0, direct_rows=0) as o:
        o.set_graph(*g.arrays())
        d, st = o.optimize(20)
    print(f"== {name}: refresh always {st['pcg_iters'][:20]}", flush=True)
    os.environ["SGO_AMG_LAG"] = "1"
    for start in (5, 8, 11, 14, 17):
        os.environ["SGO_AMG_LAG_FORCE"] = str(start)
        with capi.Optimizer(0, direct_rows=0) as o:
            o.set_graph(*g.arrays())
            print(f"-- {name}: kept from iteration {start} on", flush=True)
            d, st = o.optimize(20)
