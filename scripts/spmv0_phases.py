#!/usr/bin/env python3
"""Per-phase cycle stamps of the level-0 tile kernel (sgo_debug_spmv0_us variant 32) + microseconds per launch back to back,
per mode.  Usage: python scripts/spmv0_phases.py [C4] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_gslam_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = synth.config(name)
L = capi.lib()
L.sgo_debug_spmv0_us.restype = C.c_double
L.sgo_debug_spmv0_us.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    o.linearize()
    for mode, label in ((0, "H p"), (1, "residual pass"), (2, "Jacobi sweep")):
        L.sgo_debug_spmv0_us(o._h, mode, 0, 50)
        a = L.sgo_debug_spmv0_us(o._h, mode, 0, reps)
        ga = L.sgo_debug_spmv0_us(o._h, mode, 128, reps)
        print(f"{name} {label:14s}: {a:7.2f} us back to back, {ga:7.2f} us as hipGraph nodes", flush=True)
        sys.stderr.flush()
        L.sgo_debug_spmv0_us(o._h, mode, 32, 1)
