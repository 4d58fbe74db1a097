import os
import sys

import pytest

# the test processes run several scheduler instances per GPU: their own choice of HIP hardware queues (INTEGRATION.md section 4;
# the library does not touch the environment on load)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests._oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from tests._oracle import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return Ref()


HOOKS_LIB = os.path.join(ROOT, "dsrc_amd", "csrc", "libdsrc_gpu_hooks.so")


@pytest.fixture
def gpu_hooks():
    """dsrc_amd._lib bound to libdsrc_gpu_hooks.so for one test: the product's sources built with -DDSRC_GPU_TEST_HOOKS, the only GPU
    build in which the DSRC_GPU_* switches that FORCE a path exist (ballot ranking, the range coder's reference loop, the sort-and-replay
    front end, arena fill ...).  The product library has none of them: it chooses by itself."""
    from dsrc_amd import _lib
    if not os.path.exists(HOOKS_LIB):
        pytest.skip("libdsrc_gpu_hooks.so not built")
    old_env, old_lib = os.environ.get("DSRC_GPU_LIB"), _lib._lib
    os.environ["DSRC_GPU_LIB"] = HOOKS_LIB
    _lib._lib = None
    try:
        yield _lib
    finally:
        _lib._lib = old_lib
        if old_env is None:
            os.environ.pop("DSRC_GPU_LIB", None)
        else:
            os.environ["DSRC_GPU_LIB"] = old_env
