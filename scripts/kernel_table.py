#!/usr/bin/env python3
"""Per-kernel time table (profile mode: every launch bracketed by its own dispatch events) of optimize(iters) on a named
config: `python scripts/kernel_table.py C4r [iters] [key=value ...]` (generator overrides, e.g. init=odom)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
over = dict(a.split("=", 1) for a in sys.argv[3:])
skip = int(over.pop("skip", 1))      # Gauss-Newton iterations run before the profiled ones (skip=15: a call's late iterations)
g = synth.config(name, **over)
with capi.Optimizer(0, profile=1) as o:
    o.set_graph(*g.arrays())
    print(o.solver_description(), flush=True)
    o.optimize(skip)
    print(o.solver_description(), flush=True)
    o.profile_reset()
    done, st = o.optimize(iters)
    prof = o.kernel_profile()
tot = sum(v["ms"] for v in prof.values())
its = sum(st["pcg_iters"])
print(f"{name}: {done} GN iterations, pcg {st['pcg_iters']}, kernel time {tot:.2f} ms = {tot / max(done, 1):.2f} ms per GN iteration, "
      f"{1e3 * tot / max(its, 1):.0f} us per PCG iteration (all kernels)")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:44s} {v['launches']:7d} launches  {v['ms']:9.3f} ms  {1e3 * v['ms'] / v['launches']:8.2f} us avg  "
          f"{v['launches'] / max(its, 1):6.2f} per PCG it  {100 * v['ms'] / tot:5.1f} %")
