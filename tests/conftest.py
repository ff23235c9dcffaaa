"""pytest configuration: the ``gpu`` marker, repo-root imports and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library reads its experiment knobs (QUANTO_HIP_LARGE_CFG ...) only when this switch was set before its first call; the
# GPU tests use the knobs to force every tile configuration through the parity gates
os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: timing assertion on a real MI355X (run with -m perf; kept out of the -m gpu parity run)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests on a host without a ROCm device are skipped, not failed (hundreds of QuantoHipError would hide real CPU regressions)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a ROCm device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the real reference (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "quanto_golden.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
