import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TESTS_DIR = os.path.join(ROOT, "tests")
if TESTS_DIR not in sys.path:
    sys.path.insert(0, TESTS_DIR)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def sparse_map(seed, shape):
    """Same generator as tests/golden/make_golden.py (ReLU-like map, ~50 % zeros)."""
    rng = np.random.default_rng(seed)
    x = rng.random(shape, dtype=np.float32)
    x *= (rng.random(shape, dtype=np.float32) > 0.5)
    return x
