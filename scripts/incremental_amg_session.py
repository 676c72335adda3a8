#!/usr/bin/env python3
"""bench.py's incremental_session leg on its own (sgo_update_graph_se2 against a fresh sgo_set_graph_se2 per closure).
Usage: python scripts/incremental_amg_session.py [V] [E] [steps] [chain] [seed]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    V, E, steps, chain, seed = (a + [100000, 1000000, 12, 25, 4][len(a):])[:5]
    compare = os.environ.get("SGO_SESSION_COMPARE", "1") != "0"   # 0: without the fresh set-up beside every update
    print(json.dumps(bench.incremental_session(0, V, E, seed, steps, chain, compare=compare), indent=1))
