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
    """Outputs of the reference itself on fixed inputs (tests/golden/make_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "reference_golden.npz")
    with np.load(path, allow_pickle=False) as data:
        return {k: data[k] for k in data.files}


@pytest.fixture(scope="session")
def steane_h():
    # test/test_css_code.py:13-17
    return np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])


@pytest.fixture(scope="session")
def rm15():
    cols = np.arange(1, 16)
    h1 = np.array([(cols >> b) & 1 for b in range(4)])
    pairs = [h1[a] & h1[b] for a in range(4) for b in range(a + 1, 4)]
    return h1, np.vstack([h1] + pairs)
