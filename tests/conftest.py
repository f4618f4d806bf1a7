import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def blmm():
    """The product package (HIP path).  Fails loudly if the library is missing."""
    import bulklmm_jl_amd as b
    b.load()
    return b


@pytest.fixture(scope="session")
def gpu_ctx(blmm):
    ctx = blmm.default_context()
    return ctx
