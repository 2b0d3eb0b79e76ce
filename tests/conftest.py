import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gb():
    """The product package; building it first if the shared library is missing."""
    so = os.path.join(ROOT, "pygraphblas_amd", "libgrb_mi355x.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    import pygraphblas_amd
    return pygraphblas_amd


@pytest.fixture(scope="session")
def gpu(gb):
    info = gb.device_info()
    if not info["ok"]:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (" + info["name"] + ")")
    return info
