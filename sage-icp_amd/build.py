"""In-tree build of libsageicp_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/kernels.hip", "csrc/sort.hip", "csrc/preprocess.hip", "csrc/map_update.hip",
           "csrc/capi.hip"]
HEADERS = ["csrc/kernels.h", "csrc/sageicp_types.h", "csrc/se3_math.h", "csrc/host_map.hpp", "csrc/pipeline.hpp", "csrc/map_update.h", "csrc/metrics.hpp", "csrc/robin_order.hpp",
           "../include/sageicp.h"]
OUT = os.path.join(HERE, "libsageicp_hip.so")

# -ffp-contract=off: fp64 distances are the plain IEEE mul/add sequence the CPU evaluates, so the
# device argmin is index-exact against the oracle (see kernels.hip header).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wextra"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT, "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
