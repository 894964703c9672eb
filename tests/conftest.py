import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc   # oracle/ package — test infrastructure only
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def sage():
    import sage_icp_amd
    sage_icp_amd.lib()
    return sage_icp_amd


@pytest.fixture(scope="session")
def gpu_sage(sage):
    if sage.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    return sage
