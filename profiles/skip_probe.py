"""CPU feasibility probe for the exact "margin budget" skip (profiles/skip_probe.cpp).

    python profiles/skip_probe.py [c2|c1|c4|c5] [cold|steady|dense] [scale] [check]

Prints, per iteration, how many queries needed a search and the step norm; at the end the share of
(query, iteration) pairs searched, the number of skipped queries whose kept neighbour differed from
a full search (must be 0) and the pose difference against the oracle's plain loop.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
import sage_icp_amd  # noqa: E402,F401  (registers the hyphenated package)
syn = __import__("sage_icp_amd").synthetic if hasattr(__import__("sage_icp_amd"), "synthetic") else None


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    params = sys.argv[2] if len(sys.argv) > 2 else "cold"
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    check = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    so = "/tmp/libskip_probe.so"
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared",
                           "-o", so, os.path.join(HERE, "skip_probe.cpp")])
    lib = C.CDLL(so)
    lib.sgo_map_create.restype = C.c_void_p
    global syn
    if syn is None:
        import importlib
        syn = importlib.import_module("sage_icp_amd.synthetic")

    class OMap:
        def __init__(self, voxel):
            labels = (C.c_int * 7)(40, 44, 48, 49, 50, 70, 72)
            self.h = C.c_void_p(lib.sgo_map_create(C.c_double(voxel), C.c_double(100.0), 20, 20, labels, 7))

        def AddPoints(self, p):
            p = np.ascontiguousarray(p, dtype=np.float64)
            lib.sgo_map_add_points(self.h, p.ctypes.data_as(C.c_void_p), C.c_uint64(len(p)))

        def size(self):
            lib.sgo_map_size.restype = C.c_uint64
            return int(lib.sgo_map_size(self.h))

    w = syn.make_workload(name, lambda: OMap(syn.WORKLOADS[name]["voxel"]), scale=scale)
    prm = syn.PARAMS[params]
    scan = np.ascontiguousarray(w["scan"])
    n = len(scan)
    searched = np.zeros(500, dtype=np.uint64)
    step = np.zeros(500)
    mism = C.c_uint64(0)
    out = np.zeros(7)
    init = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
    lib.probe_skip.restype = C.c_int
    it = lib.probe_skip(w["map"].h, scan.ctypes.data_as(C.c_void_p), C.c_uint64(n), init.ctypes.data_as(C.c_void_p),
                        C.c_double(prm["max_dist"]), C.c_double(prm["kernel"]), C.c_double(prm["sem_th"]), 0, check,
                        searched.ctypes.data_as(C.c_void_p), step.ctypes.data_as(C.c_void_p), C.byref(mism),
                        out.ctypes.data_as(C.c_void_p))
    ref = np.zeros(7)
    st = oracle.Stats()
    lib.sgo_register_frame(w["map"].h, scan.ctypes.data_as(C.c_void_p), C.c_uint64(n), init.ctypes.data_as(C.c_void_p),
                           C.c_double(prm["max_dist"]), C.c_double(prm["kernel"]), C.c_double(prm["sem_th"]),
                           ref.ctypes.data_as(C.c_void_p), C.byref(st), 0)
    print(f"{name} {params} scale {scale}: {n} queries, {it} iterations (oracle: {st.iterations})")
    for k in range(it):
        if k < 12 or k % 10 == 0 or k == it - 1:
            print(f"  it {k:3d}  searched {int(searched[k]):7d} ({100.0 * searched[k] / n:5.1f} %)  step {step[k]:.3e}")
    tot = int(searched[:it].sum())
    print(f"searched {tot} of {n * it} (query, iteration) pairs = {100.0 * tot / (n * it):.2f} %;"
          f" without iteration 0: {100.0 * (tot - n) / max(1, n * (it - 1)):.2f} %")
    print(f"skipped queries whose neighbour differed from a full search: {mism.value}")
    print(f"max |pose - oracle pose| = {np.abs(out - ref).max():.3e}")


if __name__ == "__main__":
    main()
