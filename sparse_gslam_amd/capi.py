"""ctypes binding of libsgo.so (include/sgo.h) -- the only way Python reaches the optimiser.

There is no CPU fallback: if the HIP library is missing or no GPU is present, calls fail
loudly (``SgoError``).  The oracle under ``oracle/`` is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsgo.so")

SGO_MAX_ITERS = 256
SOLVER_PCG_BJ = 0
SOLVER_PCG_AMG = 1
UNIQUE_ID_BYTES = 128

# every symbol include/sgo.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "sgo_version", "sgo_default_opts", "sgo_create", "sgo_destroy", "sgo_set_graph_se2",
    "sgo_set_poses", "sgo_get_poses", "sgo_optimize_gn", "sgo_chi2", "sgo_edge_chi2",
    "sgo_num_free", "sgo_free_ids", "sgo_linearize", "sgo_hessian_apply", "sgo_solve",
    "sgo_precondition", "sgo_kernel_profile", "sgo_profile_reset", "sgo_profile_overhead_ms", "sgo_comm_unique_id",
    "sgo_comm_init", "sgo_comm_size", "sgo_shard_range", "sgo_debug_set_shard", "sgo_last_error",
    "sgo_closure_information", "sgo_plan_rows", "sgo_mfront_plan", "sgo_debug_coarse_rhs", "sgo_debug_spmv0_us",
    "sgo_solver_description", "sgo_comm_init_host", "sgo_comm_host_allgather", "sgo_debug_level0_bytes",
    "sgo_kernel_profile_samples", "sgo_update_graph_se2", "sgo_debug_lanczos",
]


class SgoError(RuntimeError):
    pass


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("solver", C.c_int32), ("pcg_tol", C.c_double),
                ("pcg_maxit", C.c_int32), ("pcg_chunk", C.c_int32), ("use_graph", C.c_int32),
                ("profile", C.c_int32), ("verbose", C.c_int32), ("direct_rows", C.c_int32),
                ("pcg_tol_cap", C.c_double), ("pcg_warm_start", C.c_int32), ("reserved", C.c_int32 * 4)]


class Stats(C.Structure):
    _fields_ = [("iters_requested", C.c_int32), ("iters_done", C.c_int32),
                ("chi2", C.c_double * (SGO_MAX_ITERS + 1)),
                ("robust_chi2", C.c_double * (SGO_MAX_ITERS + 1)),
                ("pcg_iters", C.c_int32 * SGO_MAX_ITERS),
                ("pcg_converged", C.c_int32 * SGO_MAX_ITERS),
                ("pcg_relres", C.c_double * SGO_MAX_ITERS),
                ("seconds", C.c_double * SGO_MAX_ITERS),
                ("seconds_linearize", C.c_double * SGO_MAX_ITERS),
                ("seconds_solve", C.c_double * SGO_MAX_ITERS),
                ("seconds_total", C.c_double), ("seconds_setup", C.c_double)]


class MatchWindow(C.Structure):
    """sgo_match_window (include/sgo.h): one scan-match score window."""
    _fields_ = [("x_index_offset", C.c_int32), ("y_index_offset", C.c_int32), ("scan_index", C.c_int32),
                ("scan_window", C.c_int32), ("w_size", C.c_int32), ("num_angular_perturbations", C.c_int32),
                ("resolution", C.c_double), ("angular_step", C.c_double), ("score_offset", C.c_int64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("ms", C.c_double),
                ("bytes", C.c_double)]


# sgo_host_allreduce_fn: int fn(double* buf, size_t count, void* user)
HOST_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_void_p)
HOST_ALLGATHER = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double), C.c_void_p)

_LIB = None


def build(force: bool = False) -> str:
    """Compile libsgo.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(_HERE, "..", "include", "sgo.h")]
    stale = (not os.path.exists(LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "libsgo.so"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SgoError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    d = C.POINTER(C.c_double)
    i32 = C.POINTER(C.c_int32)
    u8 = C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.sgo_version.restype = C.c_int
    L.sgo_default_opts.argtypes = [C.POINTER(Opts)]
    L.sgo_create.restype = vp
    L.sgo_create.argtypes = [C.c_int, C.POINTER(Opts)]
    L.sgo_destroy.argtypes = [vp]
    L.sgo_set_graph_se2.argtypes = [vp, C.c_int32, d, u8, C.c_int32, i32, i32, d, d, d]
    L.sgo_update_graph_se2.argtypes = [vp, C.c_int32, d, u8, C.c_int32, i32, i32, d, d, d, C.c_int32]
    L.sgo_debug_lanczos.argtypes = [vp, d, C.c_int]
    L.sgo_set_poses.argtypes = [vp, d]
    L.sgo_get_poses.argtypes = [vp, d]
    L.sgo_optimize_gn.argtypes = [vp, C.c_int32, C.POINTER(Stats)]
    L.sgo_chi2.argtypes = [vp, d, d]
    L.sgo_edge_chi2.argtypes = [vp, d]
    L.sgo_num_free.argtypes = [vp]
    L.sgo_free_ids.argtypes = [vp, i32]
    L.sgo_linearize.argtypes = [vp, d, d, d, d]
    L.sgo_hessian_apply.argtypes = [vp, d, d]
    L.sgo_solve.argtypes = [vp, d, d]
    L.sgo_precondition.argtypes = [vp, d, d]
    L.sgo_kernel_profile.argtypes = [vp, C.POINTER(KernelStat), C.c_int]
    L.sgo_kernel_profile_samples.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int]
    L.sgo_solver_description.restype = C.c_char_p
    L.sgo_solver_description.argtypes = [vp]
    L.sgo_profile_reset.argtypes = [vp]
    L.sgo_profile_overhead_ms.restype = C.c_double
    L.sgo_profile_overhead_ms.argtypes = [vp]
    L.sgo_comm_unique_id.argtypes = [vp]
    L.sgo_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.sgo_comm_size.argtypes = [vp]
    L.sgo_comm_init_host.argtypes = [vp, C.c_int, C.c_int, HOST_ALLREDUCE, vp]
    L.sgo_comm_host_allgather.argtypes = [vp, HOST_ALLGATHER]
    L.sgo_debug_level0_bytes.restype = C.c_int64
    L.sgo_debug_level0_bytes.argtypes = [vp]
    L.sgo_shard_range.restype = None
    L.sgo_shard_range.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32, i32]
    L.sgo_debug_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.sgo_closure_information.argtypes = [vp, C.c_int32, C.POINTER(MatchWindow), C.POINTER(C.c_float), C.c_int64, d, d]
    L.sgo_plan_rows.argtypes = [C.c_int32, d, u8, C.c_int32, i32, i32, C.c_int32, i32, i32, i32, i32, C.c_int32, i32, d]
    L.sgo_last_error.restype = C.c_char_p
    L.sgo_last_error.argtypes = [vp]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def default_opts() -> Opts:
    o = Opts()
    lib().sgo_default_opts(C.byref(o))
    return o


def shard_range(count: int, nranks: int, rank: int):
    """[begin, end) of the work units rank `rank` evaluates (sgo_shard_range; needs no GPU)."""
    a = C.c_int32()
    b = C.c_int32()
    lib().sgo_shard_range(count, nranks, rank, C.byref(a), C.byref(b))
    return a.value, b.value


def plan_rows(poses, fixed, ei, ej, nranks: int = 1, meas=None):
    """Host-only row plan of a graph (sgo_plan_rows; needs no GPU): dict with n, row_vertex (n,),
    tile_row_begin (ntiles + 1,), rank_row_begin (nranks + 1,).  meas: the measurements (as set_graph's) -- with them the plan is
    also right for a graph whose initial poses contradict its closures (rows ordered by spanning-tree positions)."""
    p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
    f = np.ascontiguousarray(fixed, dtype=np.uint8)
    a = np.ascontiguousarray(ei, dtype=np.int32)
    b = np.ascontiguousarray(ej, dtype=np.int32)
    V = p.shape[0]
    n = C.c_int32()
    nt = C.c_int32()
    rv = np.empty(V, dtype=np.int32)
    tb = np.empty(V + 2, dtype=np.int32)
    rb = np.empty(nranks + 1, dtype=np.int32)
    rc = lib().sgo_plan_rows(V, _dp(p), f.ctypes.data_as(C.POINTER(C.c_uint8)), a.size, _ip(a), _ip(b), nranks,
                             C.byref(n), _ip(rv), C.byref(nt), _ip(tb), tb.size, _ip(rb),
                             _dp(np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 3)) if meas is not None else None)
    if rc != 0:
        raise SgoError(f"sgo_plan_rows: rc={rc}: " + lib().sgo_last_error(None).decode())
    return dict(n=n.value, row_vertex=rv[: n.value].copy(), tile_row_begin=tb[: nt.value + 1].copy(),
                rank_row_begin=rb.copy())


MFRONT_STATS = ("n", "fronts", "levels", "max_dim", "max_own", "max_bnd", "flops", "crit_flops", "crit_panels",
                "arena_bytes", "order_kind", "targets")


def mfront_plan(poses, fixed, ei, ej, leaf: int = 0, max_crit_mflop: float = 0.0):
    """Host-only elimination plan of the multifrontal path (sgo_mfront_plan; needs no GPU): dict of MFRONT_STATS plus
    ``qualifies``, ``why``, ``elim_vertex`` (n,), ``front_of_elim`` (n,)."""
    p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
    f = np.ascontiguousarray(fixed, dtype=np.uint8)
    a = np.ascontiguousarray(ei, dtype=np.int32)
    b = np.ascontiguousarray(ej, dtype=np.int32)
    V = p.shape[0]
    st = np.zeros(12, dtype=np.int64)
    ev = np.full(V, -1, dtype=np.int32)
    fe = np.full(V, -1, dtype=np.int32)
    L = lib()
    L.sgo_mfront_plan.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.c_int32, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32)]
    rc = L.sgo_mfront_plan(V, _dp(p), f.ctypes.data_as(C.POINTER(C.c_uint8)), a.size, _ip(a), _ip(b), leaf, max_crit_mflop,
                           st.ctypes.data_as(C.POINTER(C.c_int64)), _ip(ev), _ip(fe))
    if rc not in (0, -1):
        raise SgoError(f"sgo_mfront_plan: rc={rc}: " + L.sgo_last_error(None).decode())
    out = {k: int(v) for k, v in zip(MFRONT_STATS, st)}
    out["qualifies"] = rc == 0
    out["why"] = "" if rc == 0 else L.sgo_last_error(None).decode()
    n = out["n"]
    out["elim_vertex"] = ev[:n].copy()
    out["front_of_elim"] = fe[:n].copy()
    return out


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib().sgo_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != 0:
        raise SgoError("sgo_comm_unique_id: " + lib().sgo_last_error(None).decode())
    return buf.raw


class Optimizer:
    """One ``sgo_ctx``: the device-side stand-in for the reference's pose-graph
    ``g2o::SparseOptimizer`` (src/sparse_gslam/include/graphs.h:31-40)."""

    def __init__(self, device: int = 0, **opts):
        L = lib()
        o = default_opts()
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, v)
        self._h = L.sgo_create(device, C.byref(o))
        if not self._h:
            raise SgoError("sgo_create: " + L.sgo_last_error(None).decode())
        self.V = self.E = 0
        self.last_stats = None

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            lib().sgo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc < 0 and rc != -1:
            raise SgoError(f"{what}: rc={rc}: " + lib().sgo_last_error(self._h).decode())
        return rc

    # -- multi-GPU
    def comm_init(self, nranks: int, rank: int, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        self._check(lib().sgo_comm_init(self._h, nranks, rank, C.cast(buf, C.c_void_p)), "sgo_comm_init")

    def comm_init_host(self, nranks: int, rank: int, allreduce, allgather=None):
        """Multi-GPU mode over the caller's transport (sgo_comm_init_host): `allreduce(a)` must replace the
        float64 numpy array `a` in place by its sum over all ranks (e.g. a gloo all_reduce); the optional
        `allgather(send, recv)` fills recv (nranks * len(send)) with every rank's send (sgo_comm_host_allgather)."""
        def _cb(buf, count, _user):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:   # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._host_cb = HOST_ALLREDUCE(_cb)   # keep the trampoline alive as long as the context
        self._check(lib().sgo_comm_init_host(self._h, nranks, rank, self._host_cb, None), "sgo_comm_init_host")
        if allgather is not None:
            def _gcb(send, count, recv, _user):
                try:
                    allgather(np.ctypeslib.as_array(send, shape=(count,)), np.ctypeslib.as_array(recv, shape=(count * nranks,)))
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1
            self._host_gcb = HOST_ALLGATHER(_gcb)
            self._check(lib().sgo_comm_host_allgather(self._h, self._host_gcb), "sgo_comm_host_allgather")

    def level0_bytes(self) -> int:
        """Device bytes of the level-0 structure this rank holds (sgo_debug_level0_bytes)."""
        return int(lib().sgo_debug_level0_bytes(self._h))

    def debug_set_shard(self, nranks: int, rank: int):
        self._check(lib().sgo_debug_set_shard(self._h, nranks, rank), "sgo_debug_set_shard")

    # -- graph
    def set_graph(self, poses, fixed, ei, ej, meas, info, phi):
        p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        f = np.ascontiguousarray(fixed, dtype=np.uint8)
        a = np.ascontiguousarray(ei, dtype=np.int32)
        b = np.ascontiguousarray(ej, dtype=np.int32)
        m = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 3)
        o = np.ascontiguousarray(info, dtype=np.float64).reshape(-1, 6)
        ph = np.ascontiguousarray(phi, dtype=np.float64)
        if not (f.shape[0] == p.shape[0] and a.size == b.size == m.shape[0] == o.shape[0] == ph.size):
            raise ValueError("inconsistent array sizes")
        self._check(lib().sgo_set_graph_se2(self._h, p.shape[0], _dp(p), f.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            a.size, _ip(a), _ip(b), _dp(m), _dp(o), _dp(ph)),
                    "sgo_set_graph_se2")
        self.V, self.E = p.shape[0], a.size

    def update_graph(self, poses, fixed, ei, ej, meas, info, phi, n_resident_edges: int):
        """sgo_update_graph_se2: the whole new graph + how many leading edges are the resident graph's (incremental set-up)."""
        p = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
        f = np.ascontiguousarray(fixed, dtype=np.uint8)
        a = np.ascontiguousarray(ei, dtype=np.int32)
        b = np.ascontiguousarray(ej, dtype=np.int32)
        m = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 3)
        o = np.ascontiguousarray(info, dtype=np.float64).reshape(-1, 6)
        ph = np.ascontiguousarray(phi, dtype=np.float64)
        if not (f.shape[0] == p.shape[0] and a.size == b.size == m.shape[0] == o.shape[0] == ph.size):
            raise ValueError("inconsistent array sizes")
        self._check(lib().sgo_update_graph_se2(self._h, p.shape[0], _dp(p), f.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               a.size, _ip(a), _ip(b), _dp(m), _dp(o), _dp(ph), int(n_resident_edges)),
                    "sgo_update_graph_se2")
        self.V, self.E = p.shape[0], a.size

    def set_poses(self, poses):
        p = np.ascontiguousarray(poses, dtype=np.float64).reshape(self.V, 3)
        self._check(lib().sgo_set_poses(self._h, _dp(p)), "sgo_set_poses")

    def get_poses(self):
        p = np.empty((self.V, 3))
        self._check(lib().sgo_get_poses(self._h, _dp(p)), "sgo_get_poses")
        return p

    @property
    def n_free(self):
        return self._check(lib().sgo_num_free(self._h), "sgo_num_free")

    def free_ids(self):
        out = np.empty(self.n_free, dtype=np.int32)
        self._check(lib().sgo_free_ids(self._h, _ip(out)), "sgo_free_ids")
        return out

    # -- optimisation
    def optimize(self, iters: int = 20):
        """optimize(iters) -> (iterations done, stats dict)."""
        st = Stats()
        rc = self._check(lib().sgo_optimize_gn(self._h, iters, C.byref(st)), "sgo_optimize_gn")
        d = max(st.iters_done, 0)
        stats = dict(
            rc=rc, iters_done=st.iters_done, chi2=list(st.chi2[: d + 1]),
            robust_chi2=list(st.robust_chi2[: d + 1]), pcg_iters=list(st.pcg_iters[:iters]),
            pcg_converged=list(st.pcg_converged[:iters]), pcg_relres=list(st.pcg_relres[:iters]),
            seconds=list(st.seconds[:iters]), seconds_linearize=list(st.seconds_linearize[:iters]),
            seconds_solve=list(st.seconds_solve[:iters]), seconds_total=st.seconds_total,
            seconds_setup=st.seconds_setup)
        self.last_stats = stats
        return rc, stats

    def chi2(self):
        a = C.c_double()
        b = C.c_double()
        self._check(lib().sgo_chi2(self._h, C.byref(a), C.byref(b)), "sgo_chi2")
        return a.value, b.value

    def edge_chi2(self):
        out = np.empty(self.E)
        self._check(lib().sgo_edge_chi2(self._h, _dp(out)), "sgo_edge_chi2")
        return out

    def closure_information(self, windows, scores):
        """Covariance and information of a batch of scan-match windows (sgo_closure_information).

        windows: sequence of dicts with the sgo_match_window fields except score_offset, which is
        assigned here (the windows' scores are laid out back to back in `scores`) unless given."""
        n = len(windows)
        arr = (MatchWindow * max(n, 1))()
        off = 0
        for q, w in enumerate(windows):
            for k in ("x_index_offset", "y_index_offset", "scan_index", "scan_window", "w_size",
                      "num_angular_perturbations", "resolution", "angular_step"):
                setattr(arr[q], k, w[k])
            arr[q].score_offset = w.get("score_offset", off)
            off += (2 * w["w_size"] + 1) ** 2 * (2 * w["scan_window"] + 1)
        sc = np.ascontiguousarray(scores, dtype=np.float32)
        cov = np.empty((n, 3, 3))
        info = np.empty((n, 3, 3))
        self._check(lib().sgo_closure_information(self._h, n, arr, sc.ctypes.data_as(C.POINTER(C.c_float)),
                                                  sc.size, _dp(cov), _dp(info)), "sgo_closure_information")
        return cov, info

    # -- single steps (parity tests)
    def linearize(self):
        n = self.n_free
        b = np.empty((n, 3))
        diag = np.empty((n, 3, 3))
        c = C.c_double()
        r = C.c_double()
        self._check(lib().sgo_linearize(self._h, _dp(b), _dp(diag), C.byref(c), C.byref(r)), "sgo_linearize")
        return b, diag, c.value, r.value

    def hessian_apply(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n_free, 3)
        y = np.empty_like(x)
        self._check(lib().sgo_hessian_apply(self._h, _dp(x), _dp(y)), "sgo_hessian_apply")
        return y

    def precondition(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(self.n_free, 3)
        z = np.empty_like(r)
        self._check(lib().sgo_precondition(self._h, _dp(r), _dp(z)), "sgo_precondition")
        return z

    def solve(self):
        x = np.empty((self.n_free, 3))
        rr = C.c_double()
        it = self._check(lib().sgo_solve(self._h, _dp(x), C.byref(rr)), "sgo_solve")
        return x, it, rr.value

    def lanczos(self):
        """(alpha, beta) of every PCG iteration of the last solve (needs env SGO_LANCZOS=1 at set_graph)."""
        buf = np.empty((2048, 2))
        n = self._check(lib().sgo_debug_lanczos(self._h, _dp(buf), 2048), "sgo_debug_lanczos")
        return buf[:n, 0].copy(), buf[:n, 1].copy()

    def last_error(self) -> str:
        """Text of the last error on this context (sgo_last_error)."""
        return lib().sgo_last_error(self._h).decode()

    def solver_description(self) -> str:
        """Which solver sgo_optimize_gn runs for the resident graph (sgo_solver_description)."""
        return lib().sgo_solver_description(self._h).decode()

    # -- profiling
    def kernel_profile(self, quantiles: bool = False):
        """Per-kernel launches / summed ms / algorithmic bytes (sgo_kernel_profile); quantiles=True adds the median, 10th and
        90th percentile of the single-launch durations in microseconds (sgo_kernel_profile_samples)."""
        arr = (KernelStat * 128)()
        n = self._check(lib().sgo_kernel_profile(self._h, arr, 128), "sgo_kernel_profile")
        out = {}
        buf = np.empty(16384, dtype=np.float32) if quantiles else None
        for k in range(min(n, 128)):
            if arr[k].launches:
                d = dict(launches=int(arr[k].launches), ms=arr[k].ms, bytes=arr[k].bytes)
                if quantiles:
                    m = lib().sgo_kernel_profile_samples(self._h, k, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
                    if m > 0:
                        q = np.percentile(1e3 * buf[:m].astype(np.float64), [10, 50, 90])
                        d.update(samples=int(m), p10_us=float(q[0]), median_us=float(q[1]), p90_us=float(q[2]))
                out[arr[k].name.decode()] = d
        return out

    def profile_overhead_ms(self):
        return lib().sgo_profile_overhead_ms(self._h)

    def profile_reset(self):
        self._check(lib().sgo_profile_reset(self._h), "sgo_profile_reset")
