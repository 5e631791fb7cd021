import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """A libbloomgpu context on device 0.  GPU tests fail loudly (never skip to a
    CPU path) when the library or the device is missing."""
    from bloomsearch_amd.gpu import Context
    c = Context((0,))
    yield c
    c.close()
