import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene():
    from multi_orbslam3_amd import synth
    return synth.Scene(640, 480)


@pytest.fixture(scope="session")
def small_scene():
    from multi_orbslam3_amd import synth
    return synth.Scene(320, 240, tex_size=(800, 600), px_per_m=100.0)
