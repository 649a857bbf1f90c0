import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle32():
    import numpy as np
    from oracle import Oracle
    return Oracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    import numpy as np
    from oracle import Oracle
    return Oracle(np.float64)


@pytest.fixture(scope="session")
def ctx():
    """One libpgicp context on cuda:0 -- fails loudly (no fallback) if the
    extension or the GPU is missing."""
    from pgslam_amd import icp
    c = icp.Context(0)
    yield c
    c.close()
