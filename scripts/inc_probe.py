"""Two sessions of tests/test_gpu_incremental.py run by hand with their numbers printed (round-4 debugging aid).  python scripts/inc_probe.py"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_incremental as t
for args, kw in [((3000, 12000, 5, 20), dict(seed=3)), ((6000, 30000, 4, 60), dict(seed=11, info_mode="full", phi=10.0, closures_per_step=2))]:
    try:
        worst, descs, its = t._session(*args, **kw)
        print("worst", worst)
        for d, i in zip(descs, its):
            print(i, "|", d.split("; incremental overlay: ")[-1] if "incremental overlay" in d else d.split("; last update: ")[-1])
    except AssertionError as e:
        print("ASSERT", str(e)[:600])
