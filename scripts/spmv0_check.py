#!/usr/bin/env python3
"""Tile kernel vs wave-group kernel of the level-0 product on one graph: max difference of H x."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402


def hub():
    g = synth.manhattan(400, 700, seed=3, info_mode="full")
    extra = np.arange(2, 200, dtype=np.int32)
    ei = np.concatenate([g.ei, np.full(extra.size, 1, np.int32)])
    ej = np.concatenate([g.ej, extra])
    meas = np.concatenate([g.meas, np.zeros((extra.size, 3))])
    info = np.concatenate([g.info, np.tile(g.info[0], (extra.size, 1))])
    phi = np.concatenate([g.phi, np.full(extra.size, 1.0)])
    return g.poses, g.fixed, ei, ej, meas, info, phi


graphs = {"hub": hub(), "C1": synth.config("C1", info_mode="full").arrays(), "C2": synth.config("C2", info_mode="full").arrays()}
for name, arrs in graphs.items():
    ys = []
    for mode in ("", "group"):
        os.environ["SGO_SPMV0"] = mode
        with capi.Optimizer(0, verbose=1) as o:
            o.set_graph(*arrs)
            o.linearize()
            x = np.random.default_rng(0).standard_normal((o.n_free, 3))
            ys.append(o.hessian_apply(x))
    d = np.abs(ys[0] - ys[1]).max(axis=1)
    print(name, "max diff", d.max(), "scale", np.abs(ys[1]).max(), "rows off:", np.nonzero(d > 1e-9 * np.abs(ys[1]).max())[0][:20], flush=True)
