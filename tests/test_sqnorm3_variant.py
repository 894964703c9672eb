"""The one unverified arithmetic assumption of the oracle AND the product — the association of the
sums behind Eigen's norm() / squaredNorm() (VoxelHashMap.cpp:87,111,178, Registration.cpp:79,137) — is a
build switch shared by both sides (SAGE_SQNORM3_ORDER: oracle/sage_oracle.cpp,
sage-icp_amd/csrc/sageicp_types.h).  The default (2) is what Eigen 3.4's Redux.h evaluates per call
site by derivation — it could not be RUN here.  These tests run the pinned part of
the suite on the association rounds 1-3 used (0: x^2 + (y^2 + z^2) everywhere), so that whoever first
compiles the reference with Eigen can set the default in one line and keep a green suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ, SAGE_SQNORM3_ORDER="0")
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args,
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(os.environ.get("SAGE_SQNORM3_ORDER", "2") == "0", reason="already the variant run")
def test_oracle_suite_on_the_other_association():
    """golden vectors (indices exact, sums within rounding) and the known-answer tests of the oracle
    built with x^2 + (y^2 + z^2) everywhere"""
    r = _run(["tests/test_golden_oracle.py", "tests/test_oracle_kat.py", "-m", "not gpu"], 900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("SAGE_SQNORM3_ORDER", "2") == "0", reason="already the variant run")
def test_gpu_parity_on_the_other_association():
    """the product built with the same switch against the oracle built with it: index-exact
    correspondences (golden vectors, random scenes, near-ties), pose parity on c2 scaled"""
    r = _run(["tests/test_gpu_parity.py", "-m", "gpu", "-k",
              "golden_vectors or correspondences_index_exact or near_ties or parity_c2_scaled or edge_cases"],
             1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
