"""The decision functions of sgo_optimize_gn (sparse_gslam_amd/csrc/sgo_rules.h: when the multigrid hierarchy's coarse operators are
kept, refreshed, rebuilt, re-aggregated or reverted) are pure functions of iteration counts -- every rank of a multi-GPU run must
take the same decision from the same numbers.  tests/cpp/rules_unit.cpp checks them on recorded count sequences; no GPU, no library."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decision_functions_on_recorded_count_sequences(tmp_path):
    exe = tmp_path / "rules_unit"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "sparse_gslam_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "rules_unit.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "rules ok" in out.stdout


def test_the_driver_uses_the_rules_header():
    """optimize_gn holds no copy of a rule's arithmetic: the constants appear in sgo_rules.h only."""
    src = open(os.path.join(ROOT, "sparse_gslam_amd", "csrc", "sgo_solve.cpp")).read()
    assert '#include "sgo_rules.h"' in src
    for literal in ("2 * call_best + 10", "4 * c->amg_best + 40", "85 * trial_old", "0.95 * c->amg_lag_slope"):
        assert literal not in src, literal
