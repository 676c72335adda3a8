#!/usr/bin/env python3
"""Time the level-0 product kernel on a resident graph, with the timing-experiment variants of Spmv0Args::dbg."""
import ctypes as C
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
g = synth.config(name)
L = capi.lib()
L.sgo_debug_spmv0_us.restype = C.c_double
L.sgo_debug_spmv0_us.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    o.linearize()
    import os
    reps = int(os.environ.get("SGO_PROBE_REPS", "200"))
    variants = [int(v) for v in os.environ.get("SGO_PROBE_VARIANTS", "0").split(",")]
    for variant in variants:
        t = [L.sgo_debug_spmv0_us(o._h, mode, variant, reps) for mode in (0, 1, 2)]
        print(f"variant {variant:2d}: AX {t[0]:.1f} us  RESID {t[1]:.1f} us  JACOBI {t[2]:.1f} us", flush=True)
