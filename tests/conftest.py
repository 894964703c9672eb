import os
import sys

import pytest

# The oracle (the checker) runs OpenMP over every core it sees; the tests' inputs are small, and on
# a 256-thread host the fork/join of hundreds of tiny parallel regions dominates (a test module run
# on its own took 4 minutes instead of seconds).  bench.py's cpu_baseline is not affected.
os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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


def _ensure_product_library():
    """The .so is git-ignored: build it in-tree (hipcc cross-compiles gfx950 without a GPU) when a
    fresh checkout has none or the sources are newer.  Loading still fails loudly if that fails."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_sageicp_build", os.path.join(ROOT, "sage-icp_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.needs_build():
        mod.build()
    if os.environ.get("SAGE_SQNORM3_ORDER", "2") == "0":
        # the suite on the other association of the 3-term squared norms (csrc/sageicp_types.h);
        # product and oracle both follow the variable
        mod.build_sqnorm3_variant()


@pytest.fixture(scope="session")
def sage():
    _ensure_product_library()
    import sage_icp_amd
    sage_icp_amd.lib()
    return sage_icp_amd


@pytest.fixture(scope="session")
def gpu_sage(sage):
    if sage.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    return sage


@pytest.fixture
def reference_emission_order(oracle):
    """The product's VoxelDownsample emits in the reference's tsl::robin_map bucket order (its
    default); the oracle does so in mode 1 (mode 3 adds the reference's erase-while-iterating
    sweep, which the product follows only for maps in reference-order mode,
    tests/test_reference_order_map.py — measured without effect on the poses)."""
    oracle.set_robin_order(1)
    yield
    oracle.set_robin_order(0)


@pytest.fixture(params=["full", "compact"])
def scan_form(request):
    """k_icp scans either the full fp64 records or the compact fp32 copy behind its filter; the
    library picks by frame size and voxel density, the tests force each form in turn."""
    old = os.environ.get("SAGEICP_FILTER")
    os.environ["SAGEICP_FILTER"] = "1" if request.param == "compact" else "0"
    yield request.param
    if old is None:
        os.environ.pop("SAGEICP_FILTER", None)
    else:
        os.environ["SAGEICP_FILTER"] = old
