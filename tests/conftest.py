import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "ngicp_small.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.build(ref=True)  # compiles the C restatement (and oracle/_ref when /root/reference exists)
    return orc


@pytest.fixture(scope="session")
def hip_lib():
    from direct_lidar_odometry_amd import build, nano_gicp
    build.build()
    return nano_gicp.load_library()
