#!/usr/bin/env python3
"""Round 6: what the multigrid set-up costs with its patterns made on the device (host aggregation; SGO_AMG_SETUP=device) against the
host set-up, on the hierarchy of a config at its bench start: a forced rebuild before the first solve of optimize(2), SGO_VERBOSE timing
lines on stderr.  Usage: python scripts/dev_setup_probe.py [config ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

os.environ["SGO_MFRONT"] = "0"
for cfg in (sys.argv[1:] or ["C4"]):
    g = synth.config(cfg)
    for mode in ("host", "device", "host", "device"):
        for k in ("SGO_AMG_FORCE_REBUILD", "SGO_AMG_SETUP", "SGO_VERBOSE"):
            os.environ.pop(k, None)
        with capi.Optimizer(0, direct_rows=0) as o:
            o.set_graph(*g.arrays())
            os.environ.update({"SGO_AMG_FORCE_REBUILD": "1", "SGO_AMG_SETUP": mode, "SGO_VERBOSE": "1"})
            print(f"=== {cfg} rebuild via {mode}", file=sys.stderr, flush=True)
            t = time.perf_counter()
            d, st = o.optimize(2)
            print(f"{cfg} {mode}: optimize(2) incl. the forced rebuild {1e3 * (time.perf_counter() - t):.1f} ms, pcg {st['pcg_iters'][:2]}", flush=True)
