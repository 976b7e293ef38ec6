import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    from oracle.make_golden import load
    return load(name)


@pytest.fixture(scope="session")
def ops_golden():
    return golden("ops.npz")


@pytest.fixture(scope="session")
def layers_golden():
    return golden("layers.npz")
