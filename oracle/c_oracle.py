"""ctypes binding of oracle/libsgo_oracle.so (the C++ CPU restatement).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/sgo_oracle.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libsgo_oracle.so")
    src = os.path.join(_HERE, "sgo_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsgo_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.sgo_oracle_normalize_theta.restype = C.c_double
        L.sgo_oracle_normalize_theta.argtypes = [C.c_double]
        L.sgo_oracle_se2_mul.argtypes = [_d, _d, _d]
        L.sgo_oracle_se2_inv.argtypes = [_d, _d]
        L.sgo_oracle_edges.argtypes = [C.c_int, _d, _d, _d, _d, _d, _d, _d, _d, _d, _d, _d]
        L.sgo_oracle_chi2.argtypes = [C.c_int, _d, C.c_int, _i, _i, _d, _d, _d,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.sgo_oracle_linearize.restype = C.c_int
        L.sgo_oracle_linearize.argtypes = [C.c_int, _d, _u, C.c_int, _i, _i, _d, _d, _d, C.c_int,
                                           _d, _d, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.sgo_oracle_hessian_apply.restype = C.c_int
        L.sgo_oracle_hessian_apply.argtypes = [C.c_int, _d, _u, C.c_int, _i, _i, _d, _d, _d, _d, _d]
        L.sgo_oracle_gn.restype = C.c_int
        L.sgo_oracle_gn.argtypes = [C.c_int, _d, _u, C.c_int, _i, _i, _d, _d, _d, C.c_int, C.c_int,
                                    C.c_double, C.c_int, _d, _d, _i, _d]
        L.sgo_oracle_pcg_timing.restype = C.c_int
        L.sgo_oracle_pcg_timing.argtypes = [C.c_int, _d, _u, C.c_int, _i, _i, _d, _d, _d, C.c_double, C.c_int, C.c_int, _d]
        _LIB = L
    return _LIB


def _prep(poses, fixed, ei, ej, meas, info, phi):
    return (np.ascontiguousarray(poses, dtype=np.float64),
            np.ascontiguousarray(fixed, dtype=np.uint8),
            np.ascontiguousarray(ei, dtype=np.int32), np.ascontiguousarray(ej, dtype=np.int32),
            np.ascontiguousarray(meas, dtype=np.float64),
            np.ascontiguousarray(info, dtype=np.float64),
            np.ascontiguousarray(phi, dtype=np.float64))


def normalize_theta(t: float) -> float:
    return lib().sgo_oracle_normalize_theta(float(t))


def se2_mul(a, b):
    out = np.empty(3)
    lib().sgo_oracle_se2_mul(np.ascontiguousarray(a, dtype=np.float64),
                             np.ascontiguousarray(b, dtype=np.float64), out)
    return out


def se2_inv(a):
    out = np.empty(3)
    lib().sgo_oracle_se2_inv(np.ascontiguousarray(a, dtype=np.float64), out)
    return out


def edges(xi, xj, z, info, phi):
    """Per-edge e (n,3), A, B (n,3,3), e2, rho0, rho1 for n independent tuples."""
    xi = np.ascontiguousarray(xi, dtype=np.float64).reshape(-1, 3)
    n = xi.shape[0]
    xj = np.ascontiguousarray(xj, dtype=np.float64).reshape(n, 3)
    z = np.ascontiguousarray(z, dtype=np.float64).reshape(n, 3)
    info = np.ascontiguousarray(info, dtype=np.float64).reshape(n, 6)
    phi = np.ascontiguousarray(np.broadcast_to(np.asarray(phi, dtype=np.float64), (n,)))
    e = np.empty((n, 3)); A = np.empty((n, 3, 3)); B = np.empty((n, 3, 3))
    e2 = np.empty(n); r0 = np.empty(n); r1 = np.empty(n)
    lib().sgo_oracle_edges(n, xi, xj, z, info, phi, e, A, B, e2, r0, r1)
    return e, A, B, e2, r0, r1


def chi2(poses, fixed, ei, ej, meas, info, phi):
    p, f, a, b, m, o, ph = _prep(poses, fixed, ei, ej, meas, info, phi)
    c = C.c_double(); r = C.c_double()
    lib().sgo_oracle_chi2(p.shape[0], p, a.size, a, b, m, o, ph, C.byref(c), C.byref(r))
    return c.value, r.value


def linearize(poses, fixed, ei, ej, meas, info, phi):
    """-> (b (n,3), diag (n,3,3), chi2, robust chi2)."""
    p, f, a, b, m, o, ph = _prep(poses, fixed, ei, ej, meas, info, phi)
    V = p.shape[0]
    bb = np.zeros((V, 3)); dd = np.zeros((V, 3, 3))
    c = C.c_double(); r = C.c_double()
    n = lib().sgo_oracle_linearize(V, p, f, a.size, a, b, m, o, ph, V, bb, dd, C.byref(c), C.byref(r))
    assert n >= 0
    return bb[:n].copy(), dd[:n].copy(), c.value, r.value


def hessian_apply(poses, fixed, ei, ej, meas, info, phi, x):
    p, f, a, b, m, o, ph = _prep(poses, fixed, ei, ej, meas, info, phi)
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    y = np.zeros_like(x)
    lib().sgo_oracle_hessian_apply(p.shape[0], p, f, a.size, a, b, m, o, ph, x, y)
    return y


def gauss_newton(poses, fixed, ei, ej, meas, info, phi, iters=20, solver="direct", pcg_tol=1e-10,
                 pcg_maxit=200000):
    """-> (poses, stats) like np_oracle.gauss_newton, plus stats['seconds'], ['iters_done']."""
    p, f, a, b, m, o, ph = _prep(poses, fixed, ei, ej, meas, info, phi)
    p = p.copy()
    c = np.zeros(iters + 1); r = np.zeros(iters + 1)
    k = np.zeros(max(iters, 1), dtype=np.int32); s = np.zeros(max(iters, 1))
    done = lib().sgo_oracle_gn(p.shape[0], p, f, a.size, a, b, m, o, ph, iters,
                               0 if solver == "direct" else 1, pcg_tol, pcg_maxit, c, r, k, s)
    d = max(done, 0)
    return p, dict(chi2=list(c[: d + 1]), robust_chi2=list(r[: d + 1]), pcg_iters=list(k[:d]),
                   seconds=list(s[:d]), iters_done=done)


def pcg_timing(poses, fixed, ei, ej, meas, info, phi, threads=1, pcg_tol=1e-8, pcg_maxit=200):
    """One GN iteration's cost with the block-Jacobi PCG on `threads` OpenMP threads (CPU-baseline variants
    B / C): dict(seconds_linearize, seconds_pcg, pcg_iters, converged)."""
    p, f, a, b, m, o, ph = _prep(poses, fixed, ei, ej, meas, info, phi)
    out = np.zeros(4)
    k = lib().sgo_oracle_pcg_timing(p.shape[0], p, f, a.size, a, b, m, o, ph, pcg_tol, pcg_maxit, threads, out)
    return dict(seconds_linearize=float(out[0]), seconds_pcg=float(out[1]), pcg_iters=int(k), converged=bool(out[3]))
