"""pytest configuration: the ``gpu`` marker, repo-root imports and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the real reference (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "quanto_golden.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
