import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / round-end driver)")


@pytest.fixture(scope="session")
def gsfm_ctx():
    """One libgsfm context per test session; fails loudly when the HIP library / GPU is missing."""
    from glomap_amd import _lib

    ctx = _lib.Context(-1)
    yield ctx
    ctx.close()
