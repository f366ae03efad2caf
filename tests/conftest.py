import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def raster_golden():
    return np.load(os.path.join(GOLDEN, "raster_golden.npz"))


@pytest.fixture(scope="session")
def env_golden():
    with open(os.path.join(GOLDEN, "env_golden.json")) as f:
        return json.load(f)


def unpack_masks(packed, size):
    return np.unpackbits(packed, axis=1)[:, :size * size].reshape(-1, size, size)
