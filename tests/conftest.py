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


@pytest.fixture(autouse=True)
def _default_tuning_after_each_test():
    """Tests that change the tuning of the shared default context (blmm_set_tuning: the switches that select another arithmetic
    path) get it back to the defaults afterwards, whatever they did."""
    yield
    mod = sys.modules.get("bulklmm_jl_amd.api") or sys.modules.get("bulklmm.jl_amd.api")
    ctx = getattr(mod, "_default_ctx", None) if mod else None
    if ctx is not None and getattr(ctx, "h", None):
        ctx.set_tuning("defaults", 0)
