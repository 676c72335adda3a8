#!/usr/bin/env python3
"""Per-kernel table of a bench.py JSON line (stdin): launches, total ms, average us, GB/s."""
import json
import sys

d = json.loads(sys.stdin.readline())
k = d["roofline"]["kernels"]
print(f"gn_ms {d['gn_iter_ms_median']:.2f}  pcg/GN {d['pcg_iters_per_gn_iter']:.1f}  value {d['value']/1e6:.2f} M/s")
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{n:<44s} launches {v['launches']:6d}  ms {v['ms']:8.3f}  avg_us {v['avg_us']:7.2f}  GB/s {v['GB/s']:7.1f}")
print("total kernel ms", round(sum(v["ms"] for v in k.values()), 2))
