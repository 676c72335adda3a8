"""bench.py's incremental session (the reference's flow on the C4-sized graph) with the round-5 refresh rules off (SGO_AMG_LAG=0)
and on.  python scripts/incremental_lag_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for lag in ("0", "1"):
    os.environ["SGO_AMG_LAG"] = lag
    r = bench.incremental_session(0, 100000, 1000000, 4)
    print("lag", lag, json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))}), flush=True)
