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


def need_built(path, what):
    """A built artefact a test stands on.  Missing on a box with a GPU (the driver's round-end run: everything was built before the
    snapshot travelled) or where /root/reference is present (build() makes all of them there) is a FAILURE -- a broken build must not
    go green by skipping; only a GPU-less machine without the reference sources may skip the tests that need oracle/_ref."""
    if os.path.exists(path):
        return
    if os.path.exists("/dev/kfd") or os.path.isdir("/root/reference/src"):
        pytest.fail(f"{what} is missing ({path}): run `python __graft_entry__.py` before the tests")
    pytest.skip(f"{what} not built (needs /root/reference at build time)")


@pytest.fixture(scope="session")
def oracle():
    from tests._oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from tests._oracle import Ref, have_ref
    if not have_ref():
        from tests._oracle import REF_SO
        need_built(REF_SO, "oracle/_ref (the unmodified reference)")
    return Ref()


HOOKS_LIB = os.path.join(ROOT, "dsrc_amd", "csrc", "libdsrc_gpu_hooks.so")


@pytest.fixture
def gpu_hooks():
    """dsrc_amd._lib bound to libdsrc_gpu_hooks.so for one test: the product's sources built with -DDSRC_GPU_TEST_HOOKS, the only GPU
    build in which the DSRC_GPU_* switches that FORCE a path exist (ballot ranking, the range coder's reference loop, the sort-and-replay
    front end, arena fill ...).  The product library has none of them: it chooses by itself."""
    from dsrc_amd import _lib
    need_built(HOOKS_LIB, "libdsrc_gpu_hooks.so")
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
