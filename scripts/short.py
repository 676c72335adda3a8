#!/usr/bin/env python3
"""Print the key fields of bench.py JSON lines read from stdin (sweep helper)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(f"value={d['value']/1e6:.2f}M/s gn_ms={d['gn_iter_ms_median']:.2f} lin_ms={d['linearize_ms_median']:.2f} "
          f"pcg={d['pcg_iters_per_gn_iter']:.1f} chi2={d['final_chi2']:.9g} iters={d['pcg_iters'][:8]}")
