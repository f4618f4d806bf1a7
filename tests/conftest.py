import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Beyond n = 124 the library hands the eigen-decomposition to rocSOLVER, whose first use in a process loads hundreds of
# megabytes of code objects (seconds, minutes on a cold machine).  The default GPU suite pins the in-house Jacobi so that
# it does not depend on that; BLMM_TEST_ROCSOLVER=1 runs the same tests (and test_rocsolver_eigen_path) through rocSOLVER.
if not os.environ.get("BLMM_TEST_ROCSOLVER"):
    os.environ.setdefault("BLMM_EIGEN", "jacobi")


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
