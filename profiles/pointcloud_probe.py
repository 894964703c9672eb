"""What sageicp_map_pointcloud() of a resident 1.44 M-point map costs, by destination buffer:
a fresh allocation per call (what `std::vector<Eigen::Vector4d> Pointcloud()` of the reference's
interface, and numpy's np.empty, hand over: untouched pages), the same buffer reused, and a reused
buffer registered with the HIP runtime (page-locked)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import sage_icp_amd as sage

rng = np.random.default_rng(3)
m = sage.VoxelHashMap(1.0, 100.0)
pts = rng.uniform(-60, 60, size=(3_000_000, 4))
pts[:, 2] = rng.uniform(-2, 6, len(pts))
pts[:, 3] = rng.choice([0, 40, 44, 50, 70], size=len(pts))
m.UpdateOnDevice(pts, sage.IDENTITY)
n = m.size()
assert m.resident()
L = sage.lib()
dp = C.POINTER(C.c_double)


def timed(label, make):
    ts = []
    for _ in range(12):
        out = make()
        t = time.perf_counter()
        L.sageicp_map_pointcloud(m._h, out.ctypes.data_as(dp), n)
        ts.append(time.perf_counter() - t)
    ts = sorted(ts[2:])
    print("%-48s %.2f ms (median of 10; %d points, %.1f MB)" % (label, 1e3 * ts[len(ts) // 2], n, n * 32 / 1e6))


timed("fresh buffer per call (np.empty)", lambda: np.empty((n, 4)))
keep = np.empty((n, 4))
keep[:] = 0.0
timed("one buffer reused", lambda: keep)
hip = C.CDLL("libamdhip64.so")
rc = hip.hipHostRegister(C.c_void_p(keep.ctypes.data), C.c_size_t(keep.nbytes), C.c_uint(0))
timed("one buffer reused, hipHostRegister'ed (rc %d)" % rc, lambda: keep)
hip.hipHostUnregister(C.c_void_p(keep.ctypes.data))
assert m.resident()
