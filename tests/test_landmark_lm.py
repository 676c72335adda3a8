"""Landmark graph: the g2o-compat header's host
Levenberg-Marquardt with numeric Jacobians, driven as src/sparse_gslam/src/drone.cpp:146-156 drives it, with
the reference's own rho-theta edge model (g2o_bindings/edge_se2_rhotheta.cpp:9-16, ls_extractor/utils.h:22-45
restated in tests/cpp/landmark_rhotheta.cpp), against the numpy LM oracle (oracle/np_lm_oracle.py) and its
committed golden vectors (tests/golden/lm_landmark.json, scripts/make_golden_lm.py).  No GPU involved.

Tolerances: lambda 1e-6 relative (the oracle differentiates EdgeSE2 numerically, the header uses its analytic
Jacobian: 2e-8 on lambda_0), robust chi2 1e-8 relative, on the iterations in which chi2 still moves by more
than 1e-9 relative -- beyond that LM with delta = 1e-9 central differences only accepts or rejects rounding
noise, and the two implementations' damping histories legitimately part."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
GOLD = os.path.join(ROOT, "tests", "golden")


def _significant(trace):
    out, prev = [], None
    for t in trace:
        if prev is not None and abs(prev - t["chi2"]) <= 1e-9 * t["chi2"]:
            break
        out.append(t)
        prev = t["chi2"]
    return out


def test_oracle_reproduces_its_golden_vectors():
    """The committed vectors are what the oracle computes today (guards the fixture against oracle drift)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_lm", os.path.join(ROOT, "scripts", "make_golden_lm.py"))
    gold = json.load(open(os.path.join(GOLD, "lm_landmark.json")))
    graph = open(os.path.join(GOLD, "lm_landmark_graph.txt")).read()
    spec.loader.exec_module(importlib.util.module_from_spec(spec))   # regenerates both files
    assert open(os.path.join(GOLD, "lm_landmark_graph.txt")).read() == graph
    again = json.load(open(os.path.join(GOLD, "lm_landmark.json")))
    for a, b in zip(gold["stages"], again["stages"]):
        assert a["iterations"] == b["iterations"] and abs(a["chi2"] - b["chi2"]) <= 1e-12 * a["chi2"]


def test_lm_kats():
    """Known answers of the restated pieces: transform_line / checkRhoTheta (utils.h:22-45) and the un-wrapped
    landmark theta (vertex_rhotheta.cpp:33)."""
    from oracle import np_lm_oracle as lm
    # a line at rho = 2 along +x seen after moving the frame by (1, 0): one metre closer
    assert np.allclose(lm.transform_line(np.array([2.0, 0.0]), np.array([-1.0, 0.0]), 0.0), [1.0, 0.0])
    # moving past the line flips the normal: rho stays non-negative, theta turns by pi (wrapped to (-pi, pi])
    r = lm.transform_line(np.array([2.0, 0.0]), np.array([-3.0, 0.0]), 0.0)
    assert np.allclose(r, [1.0, np.pi])
    r = lm.transform_line(np.array([1.0, 3.0]), np.array([0.0, 0.0]), 0.5)   # 3.5 > pi wraps once
    assert np.allclose(r, [1.0, 3.5 - 2 * np.pi])
    g = lm.Graph()
    g.v[7] = dict(kind="line", est=np.array([1.0, 3.0]), fixed=False)
    assert g.oplus(7, {7: g.v[7]["est"]}, np.array([0.0, 0.5]))[1] == 3.5     # NOT wrapped


def test_shim_lm_matches_the_oracle_goldens():
    subprocess.check_call(["make", "-s", "-C", CPP, "landmark_rhotheta"])
    out = subprocess.run([os.path.join(CPP, "landmark_rhotheta"), os.path.join(GOLD, "lm_landmark_graph.txt")],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    gold = json.load(open(os.path.join(GOLD, "lm_landmark.json")))
    stages, verts = [], {}
    for line in out:
        tok = line.split()
        if tok[0] == "STAGE":
            stages.append(dict(iterations=int(tok[1]), chi2=float(tok[2]), trace=[]))
        elif tok[0] == "IT":
            stages[-1]["trace"].append(dict(lam=float(tok[1]), chi2=float(tok[2]), trials=int(tok[3])))
        elif tok[0] == "V":
            verts[tok[1]] = [float(x) for x in tok[2:]]
    assert len(stages) == 2
    for s, gs in zip(stages, gold["stages"]):
        sig = _significant(gs["trace"])
        assert len(sig) >= 3 and len(s["trace"]) >= len(sig)
        for k, gt in enumerate(sig):
            st = s["trace"][k]
            assert abs(st["lam"] - gt["lam"]) <= 1e-6 * gt["lam"], (k, st, gt)
            assert abs(st["chi2"] - gt["chi2"]) <= 1e-8 * gt["chi2"], (k, st, gt)
            assert st["trials"] == gt["trials"], (k, st, gt)
        assert abs(s["chi2"] - gs["chi2"]) <= 1e-9 * gs["chi2"]
        assert 1 <= s["iterations"] <= 15
    for vid, est in gold["final"].items():
        assert np.abs(np.array(verts[vid]) - np.array(est)).max() <= 1e-6, vid


REF = "/root/reference/src"


import pytest  # noqa: E402


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_reference_binding_sources_compile_verbatim_and_reproduce_the_goldens(tmp_path):
    """The reference's own src/g2o_bindings/{edge_se2_rhotheta,vertex_rhotheta}.cpp -- and through them
    ls_extractor/utils.h -- compiled verbatim from the read-only checkout against the compat headers (the only stand-in is
    tests/ref_stub/boost/array.hpp = std::array, for two covariance helpers the path never calls), linked with the landmark
    program, which then runs the golden schedule through THEIR computeError / oplusImpl.  (src/graphs.cpp is not built:
    graphs.h pulls in pose_with_observation.h -> Cartographer + ROS message headers, libraries the image lacks; its two
    set-up functions are what tests/cpp/replay_posegraph.cpp and landmark_graph.cpp spell out.)
    Two checks: (1) their computeError against this repo's restatement on 2000 pose / line pairs: equal to rounding (1e-13);
    (2) the golden LM schedule: lambda_0 to 1e-6, every significant iterate's chi2 to 1e-5 -- the numeric Jacobians (central
    differences, delta = 1e-9) amplify the last-bit differences between two compilations of the same formula by 1e9, which is
    the 4e-7 seen in chi2 after the first iteration; the restated classes, compiled in one unit with the solver, hold 1e-8."""
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "eigen_stub"),
           "-I" + os.path.join(ROOT, "tests", "ref_stub"), "-I" + os.path.join(REF, "sparse_gslam", "include"),
           "-I" + os.path.join(REF, "ls_extractor", "include")]
    objs = []
    for name in ("edge_se2_rhotheta", "vertex_rhotheta"):
        o = str(tmp_path / (name + ".o"))
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-c", os.path.join(REF, "sparse_gslam", "src", "g2o_bindings", name + ".cpp"),
                               "-o", o] + inc)
        objs.append(o)
    chk = str(tmp_path / "ref_sources_check")
    link = ["-L" + os.path.join(ROOT, "sparse_gslam_amd", "csrc"), "-lsgo", "-L/opt/rocm/lib",
            "-Wl,-rpath," + os.path.join(ROOT, "sparse_gslam_amd", "csrc"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(["g++", "-std=c++14", "-O2", os.path.join(CPP, "ref_sources_check.cpp")] + objs + inc + link + ["-o", chk])
    r = subprocess.run([chk], capture_output=True, text=True)
    assert r.returncode == 0 and float(r.stdout.split()[0]) <= 1e-13, r.stdout + r.stderr
    exe = str(tmp_path / "landmark_ref")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-DSGO_REF_SOURCES", os.path.join(CPP, "landmark_rhotheta.cpp")] + objs + inc +
                          ["-L" + os.path.join(ROOT, "sparse_gslam_amd", "csrc"), "-lsgo", "-L/opt/rocm/lib",
                           "-Wl,-rpath," + os.path.join(ROOT, "sparse_gslam_amd", "csrc"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, os.path.join(GOLD, "lm_landmark_graph.txt")], capture_output=True, text=True, check=True).stdout.splitlines()
    gold = json.load(open(os.path.join(GOLD, "lm_landmark.json")))
    stages = []
    for line in out:
        tok = line.split()
        if tok[0] == "STAGE":
            stages.append(dict(iterations=int(tok[1]), chi2=float(tok[2]), trace=[]))
        elif tok[0] == "IT":
            stages[-1]["trace"].append(dict(lam=float(tok[1]), chi2=float(tok[2]), trials=int(tok[3])))
    assert len(stages) == 2
    for s, gs in zip(stages, gold["stages"]):
        sig = _significant(gs["trace"])
        for k, gt in enumerate(sig):
            st = s["trace"][k]
            assert abs(st["chi2"] - gt["chi2"]) <= 1e-5 * gt["chi2"], (k, st, gt)
            if k == 0:
                assert abs(st["lam"] - gt["lam"]) <= 1e-6 * gt["lam"] and st["trials"] == gt["trials"], (st, gt)
        assert abs(s["chi2"] - gs["chi2"]) <= 1e-5 * gs["chi2"]
