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
