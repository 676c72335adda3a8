#!/usr/bin/env python3
"""Builds tests/golden/ref_trajectories.npz from the keyframe trajectories the reference ships as
CARMEN result files (src/sparse_gslam/datasets/intel-lab/30pts.txt, aces/aces-30pts.txt: the output
format of log_runner.cpp:19-34).  Data only: (V,3) arrays x, y, theta, sorted by time stamp, theta as
written (unwrapped); sparse_gslam_amd.synth.trajectory_graph builds C1-sized pose graphs on them.
Run in the build container (needs /root/reference); the .npz is committed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_gslam_amd import graph_io  # noqa: E402

REF = "/root/reference/src/sparse_gslam/datasets"
out = {}
for key, rel in (("intel_lab", "intel-lab/30pts.txt"), ("aces", "aces/aces-30pts.txt")):
    P, t = graph_io.read_carmen_result(os.path.join(REF, rel))
    o = np.argsort(t, kind="stable")
    out[key] = P[o]
    out[key + "_t"] = t[o]
    print(key, out[key].shape)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_trajectories.npz"), **out)
