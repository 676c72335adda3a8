"""Pose-graph file formats either side of the optimiser.

* g2o text format (``VERTEX_SE2 id x y theta``, ``EDGE_SE2 i j dx dy dtheta o11 o12 o13 o22 o23 o33``,
  ``FIX id``): lets public pose-graph datasets (M3500, intel.g2o, ...) be fed to libsgo when a file is
  supplied.  The reference's own ``read``/``write`` members are stubs
  (src/sparse_gslam/src/g2o_bindings/edge_se2_rhotheta.cpp:18-23), so this is new I/O, not a parity
  target.
* CARMEN-style trajectory (``FLASER 0 x y theta x y theta t myhost t``): the ``<dataset>.result``
  format the reference writes (src/sparse_gslam/src/log_runner.cpp:18-23, :258-268) for the external
  ``metricEvaluator`` (datasets/eval.sh:2-3).
"""
from __future__ import annotations

import numpy as np

from .synth import Graph


def read_g2o(path: str, *, loop_phi: float = 1.0, fix_first: bool = True) -> Graph:
    """Read VERTEX_SE2 / EDGE_SE2 / FIX records.  Vertex ids are compacted to 0..V-1 in ascending id
    order.  Edges between consecutive ids are treated as odometry (no robust kernel); every other
    edge gets the DCS parameter ``loop_phi`` (pass a negative value for no kernel), which is how the
    reference attaches kernels (submap_loop_closer.cpp:214-221 vs :283)."""
    vid, vpose, fixed_ids, edges = [], [], set(), []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "VERTEX_SE2":
                vid.append(int(t[1]))
                vpose.append([float(t[2]), float(t[3]), float(t[4])])
            elif t[0] == "EDGE_SE2":
                edges.append((int(t[1]), int(t[2]), [float(v) for v in t[3:6]], [float(v) for v in t[6:12]]))
            elif t[0] == "FIX":
                fixed_ids.update(int(v) for v in t[1:])
    if not vid:
        raise ValueError(f"{path}: no VERTEX_SE2 records")
    order = np.argsort(np.array(vid), kind="stable")
    ids = np.array(vid)[order]
    index = {int(v): k for k, v in enumerate(ids)}
    poses = np.array(vpose, dtype=np.float64)[order]
    fixed = np.array([int(v) in fixed_ids for v in ids], dtype=bool)
    if fix_first and not fixed.any():
        fixed[0] = True
    E = len(edges)
    ei = np.empty(E, dtype=np.int32)
    ej = np.empty(E, dtype=np.int32)
    meas = np.empty((E, 3))
    info = np.empty((E, 6))
    phi = np.empty(E)
    for k, (a, b, m, o) in enumerate(edges):
        if a not in index or b not in index:
            raise ValueError(f"{path}: edge {a}-{b} references an unknown vertex")
        ei[k], ej[k] = index[a], index[b]
        meas[k], info[k] = m, o
        phi[k] = -1.0 if abs(index[a] - index[b]) == 1 else loop_phi
    return Graph(poses=poses, fixed=fixed, ei=ei, ej=ej, meas=meas, info=info, phi=phi,
                 meta=dict(source=path, V=len(ids), E=E, ids=ids))


def write_g2o(path: str, g: Graph, poses: np.ndarray | None = None) -> None:
    p = g.poses if poses is None else np.asarray(poses)
    with open(path, "w") as f:
        for k in range(g.V):
            f.write(f"VERTEX_SE2 {k} {float(p[k, 0])!r} {float(p[k, 1])!r} {float(p[k, 2])!r}\n")
        for k in np.flatnonzero(g.fixed):
            f.write(f"FIX {int(k)}\n")
        for k in range(g.E):
            vals = " ".join(repr(float(v)) for v in (*g.meas[k], *g.info[k]))
            f.write(f"EDGE_SE2 {int(g.ei[k])} {int(g.ej[k])} {vals}\n")


def write_carmen_result(path: str, poses: np.ndarray, times: np.ndarray | None = None) -> None:
    """One ``FLASER 0 x y theta x y theta t myhost t`` line per pose (log_runner.cpp:18-23)."""
    poses = np.asarray(poses, dtype=np.float64).reshape(-1, 3)
    t = np.arange(poses.shape[0], dtype=np.float64) if times is None else np.asarray(times, dtype=np.float64)
    with open(path, "w") as f:
        for (x, y, th), tt in zip(poses, t):
            f.write(f"FLASER 0 {x:g} {y:g} {th:g} {x:g} {y:g} {th:g} {tt:g} myhost {tt:g}\n")


def read_carmen_result(path: str) -> tuple[np.ndarray, np.ndarray]:
    P, T = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if t and t[0] == "FLASER":
                n = int(t[1])
                P.append([float(v) for v in t[2 + n: 5 + n]])
                T.append(float(t[8 + n]))
    return np.array(P), np.array(T)
