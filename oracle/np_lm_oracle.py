"""numpy restatement of the landmark-graph half of the g2o path sparse-gslam uses (SURVEY.md section 8(a) row
a12, section 8(f) rank 2): Levenberg-Marquardt over a dense system with SE(2) poses, (rho, theta) line
landmarks, pose-pose EdgeSE2 and pose-line EdgeSE2RhoTheta with NUMERIC Jacobians.

TEST INFRASTRUCTURE ONLY: imported by tests/ and scripts/make_golden_lm.py, never by the product.
Parity unpinned, as the rest of oracle/: g2o is not vendored under /root/reference and the reference has no
vectors for this path; what is restated, and from where:
  * the solver stack and call sequence: src/sparse_gslam/src/graphs.cpp:9-15 (OptimizationAlgorithmLevenberg,
    BlockSolverTraits<-1, 2>, LinearSolverEigen), src/sparse_gslam/src/drone.cpp:146-187
    (initializeOptimization / updateInitialization, push, optimize(15, online), chi2 gate, pop / discardTop);
  * the reference's own edge and vertex: src/sparse_gslam/src/g2o_bindings/edge_se2_rhotheta.cpp:9-16
    (error = z - transform_line(line, pose^-1), theta component wrapped),
    src/sparse_gslam/src/g2o_bindings/vertex_rhotheta.cpp:28-34 (additive oplus; the result of normalize_theta
    is DISCARDED there, so the landmark's theta is not wrapped),
    src/ls_extractor/include/ls_extractor/utils.h:22-45 (transform_line, checkRhoTheta);
  * g2o 2020.5.29's published algorithm (ros-gbp/libg2o-release, noetic): OptimizationAlgorithmLevenberg::solve
    -- lambda_0 = 1e-5 max diag(H) at iteration 0 of every optimize() call (also when online: `online` only
    decides whether the block structure is rebuilt or updated incrementally, never the numerics), gain ratio
    rho = (chi2 - chi2_new) / (x.(lambda x + b) + 1e-3), good step: lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)),
    nu = 2; bad step: lambda *= nu, nu *= 2, at most 10 trials, Terminate when 10 trials failed or rho == 0 --
    and BaseBinaryEdge::linearizeOplus -- central differences with delta = 1e-9 through the vertex's oplus.
"""
from __future__ import annotations

import math

import numpy as np

PI = math.pi


def normalize_theta(t: float) -> float:
    if -PI <= t < PI:
        return t
    t = t - math.floor(t / (2 * PI)) * 2 * PI
    if t >= PI:
        t -= 2 * PI
    if t < -PI:
        t += 2 * PI
    return t


def se2_inv(p):
    th = normalize_theta(-p[2])
    c, s = math.cos(th), math.sin(th)
    return np.array([c * (-p[0]) - s * (-p[1]), s * (-p[0]) + c * (-p[1]), th])


def se2_mul(a, b):
    c, s = math.cos(a[2]), math.sin(a[2])
    return np.array([a[0] + c * b[0] - s * b[1], a[1] + s * b[0] + c * b[1], normalize_theta(a[2] + b[2])])


def transform_line(rt, trans, angle):
    """ls_extractor/utils.h:33-45 (+ checkRhoTheta :22-30): the line (rho, theta) moved by (trans, angle)."""
    th = rt[1] + angle
    if th > PI:
        th -= 2 * PI
    if th < -PI:
        th += 2 * PI
    rho = rt[0] + trans[0] * math.cos(th) + trans[1] * math.sin(th)
    if rho < 0.0:
        rho = -rho
        th += PI
        if th > PI:
            th -= 2 * PI
    return np.array([rho, th])


class Graph:
    """vertices: id -> dict(kind 'pose' | 'line', est ndarray, fixed); edges: list of dict(kind 'odom' | 'obs',
    vi, vj, z ndarray, info ndarray (3x3 | 2x2))."""

    def __init__(self):
        self.v = {}
        self.e = []

    def error(self, e, est=None):
        est = est or {k: v["est"] for k, v in self.v.items()}
        xi, xj = est[e["vi"]], est[e["vj"]]
        if e["kind"] == "odom":      # EdgeSE2::computeError
            d = se2_mul(se2_inv(e["z"]), se2_mul(se2_inv(xi), xj))
            return np.array([d[0], d[1], d[2]])
        pinv = se2_inv(xi)           # EdgeSE2RhoTheta::computeError
        pred = transform_line(xj, pinv[:2], pinv[2])
        err = e["z"] - pred
        err[1] = normalize_theta(err[1])
        return err

    def oplus(self, vid, est, d):
        x = est[vid].copy()
        if self.v[vid]["kind"] == "pose":
            x[0] += d[0]
            x[1] += d[1]
            x[2] = normalize_theta(x[2] + d[2])
        else:                        # VertexRhoTheta::oplusImpl: no wrap (the wrapped value is discarded)
            x[0] += d[0]
            x[1] += d[1]
        return x

    def chi2(self, est):
        return float(sum(self.error(e, est) @ e["info"] @ self.error(e, est) for e in self.e))

    def build(self, est, order):
        """Dense normal equations at `est`: numeric Jacobians (delta = 1e-9, central) for every edge --
        EdgeSE2's analytic Jacobian agrees with them to 1e-6, which changes lambda's trace below 1e-6."""
        off, n = {}, 0
        for vid in order:
            if not self.v[vid]["fixed"]:
                off[vid] = n
                n += est[vid].size
        H = np.zeros((n, n))
        b = np.zeros(n)
        delta = 1e-9
        for e in self.e:
            err = self.error(e, est)
            J = {}
            for vid in (e["vi"], e["vj"]):
                if self.v[vid]["fixed"]:
                    continue
                dim = est[vid].size
                Jv = np.zeros((err.size, dim))
                for d in range(dim):
                    add = np.zeros(dim)
                    add[d] = delta
                    ep = dict(est)
                    ep[vid] = self.oplus(vid, est, add)
                    e1 = self.error(e, ep)
                    add[d] = -delta
                    ep[vid] = self.oplus(vid, est, add)
                    e2 = self.error(e, ep)
                    Jv[:, d] = (e1 - e2) / (2 * delta)
                J[vid] = Jv
            W = e["info"]
            for va, Ja in J.items():
                b[off[va]:off[va] + Ja.shape[1]] -= Ja.T @ W @ err
                for vb, Jb in J.items():
                    H[off[va]:off[va] + Ja.shape[1], off[vb]:off[vb] + Jb.shape[1]] += Ja.T @ W @ Jb
        return H, b, off


def levenberg(g: Graph, iterations: int):
    """SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; returns
    (iterations counted, trace [(lambda after the iteration, chi2 after, trials)])."""
    order = sorted(g.v)
    est = {k: v["est"].copy() for k, v in g.v.items()}
    lam, ni = 0.0, 2.0
    trace, done, ok = [], 0, True
    for it in range(iterations):
        if not ok:
            break
        cur = g.chi2(est)
        H, b, off = g.build(est, order)
        if it == 0:
            lam = 1e-5 * float(np.max(np.abs(np.diag(H))))
            ni = 2.0
        rho, q = 0.0, 0
        while True:
            try:
                L = np.linalg.cholesky(H + lam * np.eye(H.shape[0]))
                x = np.linalg.solve(L.T, np.linalg.solve(L, b))
                ok2 = bool(np.all(np.isfinite(x)))
            except np.linalg.LinAlgError:
                ok2, x = False, np.zeros_like(b)
            trial = dict(est)
            tmp = float("inf")
            scale = 1e-3
            if ok2:
                for vid, o in off.items():
                    trial[vid] = g.oplus(vid, est, x[o:o + est[vid].size])
                tmp = g.chi2(trial)
                scale += float(x @ (lam * x + b))
            rho = (cur - tmp) / scale
            if rho > 0 and math.isfinite(tmp) and ok2:
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                cur = tmp
                est = trial
            else:
                lam *= ni
                ni *= 2
                if not math.isfinite(lam):
                    q += 1
                    break
            q += 1
            if not (rho < 0 and q < 10):
                break
        done += 1
        trace.append((lam, cur, q))
        if q == 10 or rho == 0 or not math.isfinite(lam):
            ok = False
    for k in g.v:
        g.v[k]["est"] = est[k]
    return done, trace
