"""In-tree build of libsageicp_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

Every .hip file is its own translation unit (no device code is called across files), so the
objects are compiled in parallel, kept under build/ next to this file (git-ignored) and only
redone when their source, a header or the flags changed; the link takes a second."""
import concurrent.futures
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/kernels.hip", "csrc/sort.hip", "csrc/preprocess.hip", "csrc/map_update.hip",
           "csrc/capi_mirror.hip", "csrc/capi_run.hip", "csrc/capi.hip"]
HEADERS = ["csrc/kernels.h", "csrc/sageicp_types.h", "csrc/se3_math.h", "csrc/host_map.hpp", "csrc/pipeline.hpp", "csrc/map_update.h", "csrc/metrics.hpp", "csrc/robin_order.hpp", "csrc/capi_internal.h", "csrc/probes.h",
           "../include/sageicp.h"]
OUT = os.path.join(HERE, "libsageicp_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

# -ffp-contract=off: fp64 distances are the plain IEEE mul/add sequence the CPU evaluates, so the
# device argmin is index-exact against the oracle (see kernels.hip header).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wextra"]


def _stamp(extra):
    h = hashlib.sha1(" ".join(FLAGS + list(extra)).encode())
    for f in HEADERS:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def _prune(keep, newest=6):
    """object directories of header / flag sets that are no longer built: the newest few stay (the
    default build, the squared-norm variant, a probe build or two), the rest go — every edit of a
    header opens a new directory of 9 MB"""
    import shutil
    dirs = [os.path.join(OBJ_DIR, d) for d in os.listdir(OBJ_DIR)]
    dirs = sorted((d for d in dirs if os.path.isdir(d) and d != keep), key=os.path.getmtime, reverse=True)
    for d in dirs[newest - 1:]:
        shutil.rmtree(d, ignore_errors=True)


def build(force=False, verbose=False, out=OUT, defines=(), jobs=None):
    """defines: extra -D flags (instrumented probe builds keep their objects apart)."""
    if not force and out == OUT and not defines and not needs_build():
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-D" + d for d in defines]
    tag = _stamp(extra)
    odir = os.path.join(OBJ_DIR, tag)
    os.makedirs(odir, exist_ok=True)
    _prune(keep=odir)

    def compile_one(src):
        path = os.path.join(HERE, src)
        obj = os.path.join(odir, os.path.basename(src) + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(path):
            return obj
        cmd = [hipcc] + FLAGS + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_sqnorm3_variant(force=False, verbose=False):
    """libsageicp_hip.v0.so: the same library with SAGE_SQNORM3_ORDER=0 (sageicp_types.h) — the
    association of the 3-term squared norms rounds 1-3 used; `SAGE_SQNORM3_ORDER=0 pytest tests` runs the suite on
    it against the oracle built the same way."""
    out = os.path.join(HERE, "libsageicp_hip.v0.so")
    if not force and os.path.exists(out):
        t = os.path.getmtime(out)
        if not any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS):
            return out
    return build(force=True, verbose=verbose, out=out, defines=("SAGE_SQNORM3_ORDER=0",))


PROBE_DEFINES = ("SAGE_NN_TIMING", "SAGE_LOOP_TIMING", "SAGE_GN_TIMING", "SAGE_ICP_DELAY_PROBE", "SAGE_LOOP_INGRID")


def compile_probe_variants(jobs=None):
    """Smoke target: the device code of kernels.hip with each probe switch (csrc/probes.h and the stamps beside the loops),
    compiled and thrown away — so that the instrumented builds profiles/ relies on do not rot.  Returns {define: stderr}
    of the variants that failed (empty: all fine)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(HERE, "csrc/kernels.hip")

    def one(d):
        r = subprocess.run([hipcc] + FLAGS + ["-D" + d, "--cuda-device-only", "-c", src, "-o", os.devnull],
                           capture_output=True, text=True)
        return d, (r.stderr if r.returncode else "")

    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(PROBE_DEFINES), os.cpu_count() or 1)) as ex:
        return {d: err for d, err in ex.map(one, PROBE_DEFINES) if err}


if __name__ == "__main__":
    import sys
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a for a in sys.argv[1:] if not a.startswith("-D")]
    build(force=True, verbose=True, out=os.path.abspath(outs[0]) if outs else OUT, defines=defs)
