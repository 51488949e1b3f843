import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_ops():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ops.npz"))


@pytest.fixture(scope="session")
def golden_model():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "model.npz"))


@pytest.fixture(scope="session")
def golden_model_r3():
    """Round-3 fixtures (tests/golden/make_goldens_r3.py): softened well-conditioned cases, sixteen tiles, two heads
    at 512^2."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "model_r3.npz"))
