"""Deterministic synthetic SE(2) pose graphs (SURVEY.md section 8(d) ``manhattan``).

The reference ships no optimiser-level benchmark inputs and its real datasets are
not vendored (src/sparse_gslam/datasets/download.sh:5-9), so BASELINE.json's
configs are instantiated by this generator.  The graph has the shape the
reference's pose graph has (src/sparse_gslam/src/submap_loop_closer.cpp:205-288,
src/sparse_gslam/src/drone.cpp:54-80): dense ids from 0, vertex 0 fixed, an
odometry chain (i, i+1) without robust kernel, loop-closure edges carrying a DCS
kernel, full 3x3 information matrices, and an initial guess obtained by chaining
measurements (slc.cpp:219).

Arrays follow include/sgo.h: poses (V,3), fixed (V,), ei/ej (E,), meas (E,3),
info (E,6) upper triangle [o11,o12,o13,o22,o23,o33], phi (E,) (<0: no kernel).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

SIGMA_XY = 0.05
SIGMA_TH = 0.02

# BASELINE.json configs -> generator arguments
CONFIGS = {
    "C1": dict(V=1_000, E=1_100, seed=1, p_random=0.0),
    "C2": dict(V=10_000, E=40_000, seed=2, p_random=0.0),
    # C3 (mit-killian end-to-end) is blocked: the dataset is not vendored and the front-end is out of scope.  "C3s" is a
    # stand-in of its pose graph's SIZE -- the keyframe graph the reference ends the run with has about 5 489 poses and
    # 7 629 edges -- with that dataset's DCS delta (src/sparse_gslam/datasets/mit-killian/slam.yaml:38: dcs_phi 0.75) and full
    # information matrices: closures everywhere, too many for the single-launch direct path's separator block, too few
    # rows to fill the chip -- the launch-bound regime between the two solvers
    "C3s": dict(V=5_489, E=7_629, seed=3, p_random=0.0, phi=0.75, info_mode="full"),
    "C4": dict(V=100_000, E=1_000_000, seed=4, p_random=0.0),
    "C4r": dict(V=100_000, E=1_000_000, seed=4, p_random=0.05),
    "C5": dict(V=1_000_000, E=10_000_000, seed=5, p_random=0.05),
}


@dataclass
class Graph:
    poses: np.ndarray          # (V,3) initial guess
    fixed: np.ndarray          # (V,) bool
    ei: np.ndarray             # (E,) int32
    ej: np.ndarray             # (E,) int32
    meas: np.ndarray           # (E,3)
    info: np.ndarray           # (E,6)
    phi: np.ndarray            # (E,) f64, <0 = no kernel
    truth: np.ndarray | None = None
    meta: dict = field(default_factory=dict)

    @property
    def V(self) -> int:
        return int(self.poses.shape[0])

    @property
    def E(self) -> int:
        return int(self.ei.shape[0])

    def arrays(self):
        return (self.poses, self.fixed, self.ei, self.ej, self.meas, self.info, self.phi)

    def subset(self, mask) -> "Graph":
        """Same vertices, edges selected by ``mask`` (edge shard for one rank)."""
        return Graph(self.poses, self.fixed, self.ei[mask], self.ej[mask], self.meas[mask],
                     self.info[mask], self.phi[mask], self.truth, dict(self.meta))


def _wrap(t):
    """[-pi, pi) wrap, same branch structure as g2o::normalize_theta."""
    t = np.asarray(t, dtype=np.float64)
    u = t - np.floor(t / (2 * np.pi)) * (2 * np.pi)
    u = np.where(u >= np.pi, u - 2 * np.pi, u)
    u = np.where(u < -np.pi, u + 2 * np.pi, u)
    return np.where((t >= -np.pi) & (t < np.pi), t, u)


def _rel(pi, pj):
    """Xi^-1 * Xj for (n,3) arrays."""
    c, s = np.cos(pi[:, 2]), np.sin(pi[:, 2])
    dx, dy = pj[:, 0] - pi[:, 0], pj[:, 1] - pi[:, 1]
    return np.stack([c * dx + s * dy, -s * dx + c * dy, _wrap(pj[:, 2] - pi[:, 2])], axis=1)


def _walk(V: int, rng: np.random.Generator) -> np.ndarray:
    """Unit-grid random walk in a ceil(2*sqrt(V))-side box; turn +-90deg w.p. 0.3."""
    L = int(math.ceil(2.0 * math.sqrt(V)))
    turn_u = rng.random(V)
    turn_s = rng.integers(0, 2, size=V)
    dxs = (1, 0, -1, 0)
    dys = (0, 1, 0, -1)
    xs = np.empty(V, dtype=np.int64)
    ys = np.empty(V, dtype=np.int64)
    hs = np.empty(V, dtype=np.int64)
    x = y = L // 2
    h = 0
    xs[0], ys[0], hs[0] = x, y, h
    for k in range(1, V):
        if turn_u[k] < 0.3:
            h = (h + (1 if turn_s[k] else 3)) & 3
        for _ in range(4):  # stay inside the box: keep turning left until the step is legal
            nx, ny = x + dxs[h], y + dys[h]
            if 0 <= nx <= L and 0 <= ny <= L:
                break
            h = (h + 1) & 3
        x, y = nx, ny
        xs[k], ys[k], hs[k] = x, y, h
    th = _wrap(hs.astype(np.float64) * (np.pi / 2))
    return np.stack([xs.astype(np.float64), ys.astype(np.float64), th], axis=1)


def _closure_candidates(truth: np.ndarray, need: int, radius: float = 2.0, min_sep: int = 10):
    """Pairs (i<j) with |t_i - t_j| <= radius and j - i > min_sep, in lexicographic order."""
    from scipy.spatial import cKDTree

    tree = cKDTree(truth[:, :2])
    r = radius
    while True:
        pairs = tree.query_pairs(r, output_type="ndarray")
        if pairs.size:
            lo = np.minimum(pairs[:, 0], pairs[:, 1])
            hi = np.maximum(pairs[:, 0], pairs[:, 1])
            keep = (hi - lo) > min_sep
            lo, hi = lo[keep], hi[keep]
        else:
            lo = hi = np.empty(0, dtype=np.int64)
        if lo.size >= need or r > 64 * radius:
            break
        r *= 1.5
    order = np.lexsort((hi, lo))  # deterministic order independent of the kd-tree traversal
    return lo[order], hi[order], r


def manhattan(V: int, E: int, seed: int, p_random: float = 0.0, *, info_mode: str = "diag",
              phi: float = 1.0, init: str = "incremental", tail: int = 200,
              sigma_xy: float = SIGMA_XY, sigma_th: float = SIGMA_TH) -> Graph:
    """``manhattan(V, E, seed, p_random)`` of SURVEY.md section 8(d).

    info_mode: "diag"  Omega = diag(1/sigma^2) (stored as the 6 upper entries);
               "full"  Omega = Q diag(1/sigma^2) Q^T with a random rotation Q per edge, and
                       noise drawn from the matching covariance (exercises off-diagonals).
    init:      "incremental" (default) the state the reference's optimize(20) call sees
                       (slc.cpp:205-224, 286-287): every pose but the last ``tail`` ones is
                       already near the optimum of the previous optimisation (ground truth
                       + N(0, (0.02 m, 0.02 m, 0.005 rad))), and the newest ``tail`` poses are
                       chained from their predecessor's estimate through the odometry
                       measurements (slc.cpp:219);
               "odom"  the whole trajectory dead-reckoned from vertex 0 (undamped GN + DCS
                       does not converge from here beyond a few hundred poses -- kept for
                       small cases and failure-path tests);
               "truth" ground truth (chi2 then measures the noise only).
    """
    if E < V - 1:
        raise ValueError("E must be >= V-1 (the odometry chain)")
    rng = np.random.default_rng(seed)
    truth = _walk(V, rng)
    return _build(truth, E, rng, seed, p_random, info_mode, phi, init, tail, sigma_xy, sigma_th, 2.0, 10)


def trajectory_graph(truth: np.ndarray, E: int, seed: int, *, info_mode: str = "full", phi: float = 10.0,
                     init: str = "incremental", tail: int = 200, sigma_xy: float = SIGMA_XY,
                     sigma_th: float = SIGMA_TH, closure_radius: float = 1.0, min_sep: int = 40) -> Graph:
    """The same construction on a GIVEN trajectory (e.g. the intel-lab keyframe trajectory the
    reference ships as a result file; fixture: reference_trajectory below): odometry chain plus
    ``E - (V-1)`` closures between poses that revisit the same place (within ``closure_radius``
    metres, at least ``min_sep`` keyframes apart), DCS delta ``phi`` on the closures
    (datasets/intel-lab/slam-11.yaml:38 uses 10)."""
    truth = np.asarray(truth, dtype=np.float64).copy()
    truth[:, 2] = _wrap(truth[:, 2])
    if E < truth.shape[0] - 1:
        raise ValueError("E must be >= V-1 (the odometry chain)")
    rng = np.random.default_rng(seed)
    return _build(truth, E, rng, seed, 0.0, info_mode, phi, init, tail, sigma_xy, sigma_th, closure_radius, min_sep)


def _build(truth, E, rng, seed, p_random, info_mode, phi, init, tail, sigma_xy, sigma_th, radius, min_sep) -> Graph:
    V = truth.shape[0]
    n_close = E - (V - 1)
    n_rand = int(round(E * p_random))
    n_rand = min(n_rand, n_close)
    n_local = n_close - n_rand

    oi = np.arange(V - 1, dtype=np.int64)
    oj = oi + 1
    li = lj = np.empty(0, dtype=np.int64)
    r_used = 0.0
    if n_local > 0:
        ci, cj, r_used = _closure_candidates(truth, n_local, radius, min_sep)
        if ci.size < n_local:  # not enough revisits: top up with random pairs
            n_rand += n_local - ci.size
            n_local = ci.size
        pick = np.sort(rng.choice(ci.size, size=n_local, replace=False))
        li, lj = ci[pick], cj[pick]
    if n_rand > 0:
        a = rng.integers(0, V, size=n_rand)
        b = (a + 1 + rng.integers(0, V - 1, size=n_rand)) % V
        ri, rj = np.minimum(a, b), np.maximum(a, b)
    else:
        ri = rj = np.empty(0, dtype=np.int64)
    ei = np.concatenate([oi, li, ri])
    ej = np.concatenate([oj, lj, rj])
    Etot = ei.size

    rel = _rel(truth[ei], truth[ej])
    sig = np.array([sigma_xy, sigma_xy, sigma_th])
    noise = rng.standard_normal((Etot, 3)) * sig
    # information matrices of noise-free graphs (sigma = 0) use the nominal sigmas
    isig = np.array([sigma_xy if sigma_xy > 0 else SIGMA_XY, sigma_xy if sigma_xy > 0 else SIGMA_XY,
                     sigma_th if sigma_th > 0 else SIGMA_TH])
    info = np.zeros((Etot, 6))
    if info_mode == "diag":
        info[:, 0] = info[:, 3] = 1.0 / isig[0]**2
        info[:, 5] = 1.0 / isig[2]**2
    elif info_mode == "full":
        # random rotation Q = Rz(a) Ry(b) Rx(c), small b,c so that theta stays the stiff axis
        a = rng.uniform(-np.pi, np.pi, Etot)
        b = rng.uniform(-0.2, 0.2, Etot)
        c = rng.uniform(-0.2, 0.2, Etot)
        Q = _euler(a, b, c)
        noise = np.einsum("nij,nj->ni", Q, noise)
        O = np.einsum("nij,j,nkj->nik", Q, 1.0 / isig**2, Q)
        info[:, 0], info[:, 1], info[:, 2] = O[:, 0, 0], O[:, 0, 1], O[:, 0, 2]
        info[:, 3], info[:, 4], info[:, 5] = O[:, 1, 1], O[:, 1, 2], O[:, 2, 2]
    else:
        raise ValueError(info_mode)
    meas = rel + noise
    meas[:, 2] = _wrap(meas[:, 2])

    ph = np.full(Etot, -1.0)
    ph[V - 1:] = phi

    if init in ("odom", "incremental"):
        if init == "odom":
            poses = np.empty_like(truth)
            poses[0] = truth[0]
            start = 0
        else:
            poses = truth + rng.standard_normal(truth.shape) * np.array([0.02, 0.02, 0.005])
            poses[:, 2] = _wrap(poses[:, 2])
            poses[0] = truth[0]
            start = max(0, V - 1 - int(tail))
        x, y, t = poses[start]
        m = meas[: V - 1]
        for k in range(start, V - 1):
            c, s = math.cos(t), math.sin(t)
            x, y = x + c * m[k, 0] - s * m[k, 1], y + s * m[k, 0] + c * m[k, 1]
            t = t + m[k, 2]
            t = float(_wrap(t))
            poses[k + 1] = (x, y, t)
    elif init == "truth":
        poses = truth.copy()
    else:
        raise ValueError(init)

    fixed = np.zeros(V, dtype=bool)
    fixed[0] = True
    return Graph(poses=poses, fixed=fixed, ei=ei.astype(np.int32), ej=ej.astype(np.int32),
                 meas=meas, info=info, phi=ph, truth=truth,
                 meta=dict(V=V, E=Etot, seed=seed, p_random=p_random, info_mode=info_mode,
                           n_odom=V - 1, n_local=int(n_local), n_random=int(n_rand),
                           closure_radius=float(r_used), init=init, tail=int(tail)))


def chain_init(poses: np.ndarray, meas_odom: np.ndarray, first: int, last: int) -> None:
    """Initial guess of the appended poses first..last (inclusive), in place: each chained from its predecessor's CURRENT
    estimate through the odometry measurement (slc.cpp:219: pose->setEstimate(prev->estimate() * edge->measurement())).
    meas_odom[k] is the measurement of edge (k, k + 1)."""
    x, y, t = poses[first - 1]
    for k in range(first - 1, last):
        c, s = math.cos(t), math.sin(t)
        m = meas_odom[k]
        x, y = x + c * m[0] - s * m[1], y + s * m[0] + c * m[1]
        t = float(_wrap(t + m[2]))
        poses[k + 1] = (x, y, t)


def append_session(V0: int, E0: int, steps: int, chain: int, seed: int, *, info_mode: str = "diag", phi: float = 1.0,
                   closures_per_step: int = 1):
    """The reference's usage pattern on a graph of ANY size (slc.cpp:205-226, :272-287): a resident graph of about V0 poses /
    E0 edges, then `steps` accepted loop closures, each appending the `chain` poses driven since the last one (with their
    odometry edges) and `closures_per_step` closure edges between one of the new poses and an old pose nearby.

    One manhattan world of V0 + steps * chain poses is generated; the base is its sub-graph on the first V0 poses (edges in
    the generator's order), the steps take the following chunks.  Returns (base Graph -- poses near the optimum, as after
    the previous optimize(20) --, list of steps, full Graph): a step is a dict with V (poses after it) and the appended
    ei / ej / meas / info / phi; the appended poses' initial guess is chain_init() from the estimates at that time."""
    Vt = V0 + steps * chain
    g = manhattan(Vt, int(round(E0 * (Vt / V0))), seed, 0.0, info_mode=info_mode, phi=phi, init="incremental", tail=0)
    keep = (g.ei < V0) & (g.ej < V0)
    base = Graph(g.poses[:V0].copy(), g.fixed[:V0].copy(), g.ei[keep], g.ej[keep], g.meas[keep], g.info[keep], g.phi[keep],
                 g.truth[:V0], dict(g.meta, V=V0, E=int(keep.sum()), init="incremental", tail=0))
    out = []
    lo, hi = np.minimum(g.ei, g.ej), np.maximum(g.ei, g.ej)
    is_odom = np.arange(g.E) < Vt - 1
    for st in range(steps):
        a, b = V0 + st * chain, V0 + (st + 1) * chain          # the appended poses [a, b)
        odo = np.arange(a - 1, b - 1)                           # edges (a-1, a) ... (b-2, b-1)
        cand = np.flatnonzero(~is_odom & (hi >= a) & (hi < b) & (lo < a))
        # the closure hangs on a pose in the middle of the chunk when one exists (slc.cpp:279: poses[mid]), else the nearest
        mid = (a + b) // 2
        cand = cand[np.argsort(np.abs(hi[cand] - mid), kind="stable")][:closures_per_step]
        idx = np.concatenate([odo, np.sort(cand)])
        out.append(dict(V=b, ei=g.ei[idx], ej=g.ej[idx], meas=g.meas[idx], info=g.info[idx], phi=g.phi[idx],
                        closures=int(cand.size)))
    return base, out, g


def _euler(a, b, c):
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    Q = np.empty(a.shape + (3, 3))
    Q[:, 0, 0] = ca * cb
    Q[:, 0, 1] = ca * sb * sc - sa * cc
    Q[:, 0, 2] = ca * sb * cc + sa * sc
    Q[:, 1, 0] = sa * cb
    Q[:, 1, 1] = sa * sb * sc + ca * cc
    Q[:, 1, 2] = sa * sb * cc - ca * sc
    Q[:, 2, 0] = -sb
    Q[:, 2, 1] = cb * sc
    Q[:, 2, 2] = cb * cc
    return Q


TRAJECTORY_FILE = None   # set by the caller (tests/conftest.py, bench.py): the package itself ships no data and reads nothing of tests/


def reference_trajectory(name: str, path: str | None = None) -> np.ndarray:
    """Keyframe trajectory (V,3) of one of the reference's shipped result files
    (src/sparse_gslam/datasets/<name>/*30pts.txt, CARMEN FLASER lines sorted by time stamp), from a fixture made by
    scripts/make_traj_fixture.py.  The file is the CALLER's: `path`, else synth.TRAJECTORY_FILE, else $SGO_REF_TRAJECTORIES."""
    import os

    path = path or TRAJECTORY_FILE or os.environ.get("SGO_REF_TRAJECTORIES")
    if not path:
        raise FileNotFoundError("reference_trajectory: no trajectory fixture given (argument `path`, synth.TRAJECTORY_FILE or "
                                "$SGO_REF_TRAJECTORIES; the repository keeps one at tests/golden/ref_trajectories.npz)")
    with np.load(path) as z:
        return z[name].copy()


def config(name: str, **overrides) -> Graph:
    """Instantiate one of BASELINE.json's synthetic configs (C1, C2, C3s, C4, C4r, C5), or "C1i" /
    "C1a": the C1-sized graph on the reference's own intel-lab / aces keyframe trajectory."""
    if name in ("C1i", "C1a"):
        truth = reference_trajectory("intel_lab" if name == "C1i" else "aces")
        kw = dict(E=truth.shape[0] - 1 + (60 if name == "C1i" else 30), seed=11)
        kw.update(overrides)
        return trajectory_graph(truth, **kw)
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return manhattan(**kw)
