import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the reference's shipped keyframe trajectories (C1i / C1a): a fixture of the TEST tree, handed to the generator by its caller
    from sparse_gslam_amd import synth
    synth.TRAJECTORY_FILE = os.path.join(ROOT, "tests", "golden", "ref_trajectories.npz")


@pytest.fixture(scope="session")
def sgo_lib():
    from sparse_gslam_amd import capi
    return capi.lib()
