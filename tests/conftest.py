import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    o = graft.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def hip(pkg):
    """The product library on a GPU box: loads it and fails loudly when it is not built."""
    pkg.load_library()
    return pkg


def so3_exp_np(w):
    w = np.asarray(w, float)
    a = np.linalg.norm(w)
    if a < 1e-12:
        return np.eye(3)
    k = w / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(a) * np.eye(3) + (1 - np.cos(a)) * np.outer(k, k) + np.sin(a) * K
