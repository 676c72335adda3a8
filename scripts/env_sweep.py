#!/usr/bin/env python3
"""optimize(20) on a named config under a list of environment settings:
python scripts/env_sweep.py C4r "" "SGO_AMG_COARSEST=900" "SGO_AMG_FCG2_DEPTH=2 SGO_AMG_COARSEST=900" ..."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

name = sys.argv[1]
g = synth.config(name)
for setting in sys.argv[2:]:
    kv = dict(s.split("=", 1) for s in setting.split())
    for k, v in kv.items():
        os.environ[k] = v
    try:
        with capi.Optimizer(0) as o:
            t = time.perf_counter()
            o.set_graph(*g.arrays())
            ts = time.perf_counter() - t
            t = time.perf_counter()
            done, st = o.optimize(20)
            to = time.perf_counter() - t
            desc = o.solver_description()
        gn = [1e3 * x for x in st["seconds"][:done]]
        print(f"[{setting or 'default'}] done={done} set_graph {1e3 * ts:.0f} ms, optimize(20) {1e3 * to:.1f} ms, GN median "
              f"{statistics.median(gn):.2f} ms, pcg {st['pcg_iters'][:done]}, chi2 {st['chi2'][done]:.9e}\n    {desc[:300]}", flush=True)
    except Exception as e:
        print(f"[{setting}] failed: {e}", flush=True)
    for k in kv:
        del os.environ[k]
