#!/usr/bin/env python3
"""Golden vectors for the landmark graph (Levenberg-Marquardt, numeric Jacobians, the reference's rho-theta edge):
writes tests/golden/lm_landmark_graph.txt (the graph, read by tests/cpp/landmark_rhotheta.cpp) and
tests/golden/lm_landmark.json (lambda / chi2 / trials per iteration and the final estimates from
oracle/np_lm_oracle.py) for the two-stage call sequence of src/sparse_gslam/src/drone.cpp:146-156:
    initializeOptimization(); push(); optimize(15, false)      [stage 0]
    + new pose, new odometry edge, new observations; updateInitialization(); push(); optimize(15, true)   [stage 1]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_lm_oracle as lm  # noqa: E402

rng = np.random.default_rng(7)
NP, NL = 9, 5
truth = [np.array([0.6 * k, 0.2 * np.sin(0.7 * k), 0.15 * k]) for k in range(NP)]
ltruth = [np.array(v) for v in ([4.0, 0.3], [6.0, 1.7], [3.0, -2.0], [8.0, 2.9], [5.0, -0.9])]
odom_cov = np.array([[0.01, 0.001, 0], [0.001, 0.02, 0], [0, 0, 0.005]])
obs_cov = np.array([[0.004, 0.0005], [0.0005, 0.002]])
lines = []   # text form of the graph
g = lm.Graph()


def add_pose(k, stage):
    init = truth[k].copy()
    if k > 0:
        init += rng.normal(0, [0.05, 0.05, 0.03])
    g.v[k] = dict(kind="pose", est=init, fixed=(k == 0))
    lines.append(f"POSE {stage} {k} {float(init[0])!r} {float(init[1])!r} {float(init[2])!r} {int(k == 0)}")


def add_odom(k, stage):
    z = lm.se2_mul(lm.se2_inv(truth[k]), truth[k + 1]) + rng.multivariate_normal(np.zeros(3), 0.2 * odom_cov)
    info = np.linalg.inv(odom_cov)
    g.e.append(dict(kind="odom", vi=k, vj=k + 1, z=z, info=info))
    u = info[np.triu_indices(3)]
    lines.append(f"ODOM {stage} {k} {k + 1} " + " ".join(repr(float(v)) for v in list(z) + list(u)))


def add_obs(k, j, stage):
    pinv = lm.se2_inv(truth[k])
    z = lm.transform_line(ltruth[j], pinv[:2], pinv[2]) + rng.multivariate_normal(np.zeros(2), 0.5 * obs_cov)
    info = np.linalg.inv(obs_cov)
    g.e.append(dict(kind="obs", vi=k, vj=10_000_000 + j, z=z, info=info))
    lines.append(f"OBS {stage} {k} {10_000_000 + j} " + " ".join(repr(float(v)) for v in list(z) + [info[0, 0], info[0, 1], info[1, 1]]))


for k in range(NP - 1):
    add_pose(k, 0)
for k in range(NP - 2):
    add_odom(k, 0)
for j in range(NL):
    init = ltruth[j] + rng.normal(0, [0.07, 0.04])
    g.v[10_000_000 + j] = dict(kind="line", est=init, fixed=False)
    lines.append(f"LINE 0 {10_000_000 + j} {float(init[0])!r} {float(init[1])!r}")
for k in range(NP - 1):
    for j in range(NL):
        if (k + j) % 2 == 0:
            add_obs(k, j, 0)
out = {"stages": []}
done, trace = lm.levenberg(g, 15)
out["stages"].append(dict(iterations=done, trace=[dict(lam=a, chi2=b, trials=c) for a, b, c in trace],
                          chi2=g.chi2({k: v["est"] for k, v in g.v.items()})))
add_pose(NP - 1, 1)
add_odom(NP - 2, 1)
for j in (0, 2, 3):
    add_obs(NP - 1, j, 1)
done, trace = lm.levenberg(g, 15)
out["stages"].append(dict(iterations=done, trace=[dict(lam=a, chi2=b, trials=c) for a, b, c in trace],
                          chi2=g.chi2({k: v["est"] for k, v in g.v.items()})))
out["final"] = {str(k): [float(x) for x in v["est"]] for k, v in g.v.items()}
gold = os.path.join(ROOT, "tests", "golden")
open(os.path.join(gold, "lm_landmark_graph.txt"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(gold, "lm_landmark.json"), "w"), indent=1)
for s in out["stages"]:
    print(s["iterations"], [round(t["lam"], 6) for t in s["trace"]][:6], s["chi2"])
