"""The C++ header shim (sage-icp_amd/shim/sage_icp/core/*.hpp) keeps the reference's
VoxelHashMap / RegisterFrame surface on top of the C ABI.  Eigen and Sophus are not in this
image, so the shim is type-checked and run against minimal stand-ins of the handful of members
it touches (tests/shim_stubs/) — a test of OUR header, not a build of the reference."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "shim_user")
    lib_dir = os.path.join(ROOT, "sage-icp_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror",
           "-I", os.path.join(ROOT, "tests", "shim_stubs"),
           "-I", os.path.join(ROOT, "sage-icp_amd", "shim"),
           "-I", os.path.join(ROOT, "sage-icp_amd", "shim_preprocessing"),   # opt-in
           "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "shim_stubs", "shim_user.cpp"),
           "-L", lib_dir, "-l:libsageicp_hip.so", "-Wl,-rpath," + lib_dir, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_shim_compiles_and_runs_host_side(tmp_path, sage):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "shim ok" in out.stdout


@pytest.mark.gpu
def test_shim_runs_on_gpu(tmp_path, gpu_sage):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "planted pose recovered" in out.stdout      # RegisterFrame's answer is checked in the program
