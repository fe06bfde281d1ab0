"""pytest configuration: markers and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def lexsorted(rows):
    """Rows as a canonical, order-free set (sorted by every column, last column fastest)."""
    rows = np.asarray(rows, dtype=np.float64)
    if rows.size == 0:
        return rows
    keys = tuple(rows[:, i] for i in range(rows.shape[1] - 1, -1, -1))
    return rows[np.lexsort(keys)]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
