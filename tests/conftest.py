import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure) -- builds oracle/liborbslam_oracle.so on first use"""
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def frames640():
    from orbslam_in_practice_amd import synth
    return synth.make_stream(7, 3, 640, 480)
