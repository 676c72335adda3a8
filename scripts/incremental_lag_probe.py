import os, sys, json
sys.path.insert(0, "/root/repo")
import bench
for lag in ("0", "1"):
    os.environ["SGO_AMG_LAG"] = lag
    r = bench.incremental_session(0, 100000, 1000000, 4)
    print("lag", lag, json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (list, dict))}), flush=True)
