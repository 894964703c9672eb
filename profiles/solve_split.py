"""Where the solving wave's ~3 us go (-DSAGE_LOOP_TIMING build): assemble + LDL^T | SE3 exp | composition + rotation matrix | norm | rest.
    python profiles/solve_split.py [workload c1]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

sage.LIB_PATH = os.environ.get("LOOP_LIB", os.path.join(os.path.dirname(sage.LIB_PATH), "_probe", "libsageicp_looptiming.so"))
from sage_icp_amd import synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
p = syn.PARAMS["cold"]
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
f = sage.Frame(w["map"], w["scan"])
os.environ["SAGEICP_LOOP"] = "2"
sage.set_counting(False)
for _ in range(3):
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
IT, WG = 32, 2048
wg = np.zeros((IT, WG, 4), dtype=np.uint64)
sv = np.zeros((IT, 4), dtype=np.uint64)
s2 = np.zeros((IT, 4), dtype=np.uint64)
sage.lib().sageicp_debug_loop_times(wg.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p))
sage.lib().sageicp_debug_loop_solver2(s2.ctypes.data_as(C.c_void_p))
sv = sv.astype(np.float64) / 100.0
s2 = s2.astype(np.float64) / 100.0
n = min(IT, st.iterations) - 1
r = slice(3, n)
print("%s: mean over iterations 3..%d, us:" % (name, n - 1))
print("   counts complete -> sums in fp64          %.2f" % (sv[r, 1] - sv[r, 0]).mean())
print("   assemble + pivoted LDL^T + substitutions %.2f" % (s2[r, 0] - sv[r, 1]).mean())
print("   SE3 exp (sqrt, sincos, divisions)        %.2f" % (s2[r, 1] - s2[r, 0]).mean())
print("   composition (2 lanes) + rotation matrix  %.2f" % (s2[r, 2] - s2[r, 1]).mean())
print("   norm of the step (sqrt)                  %.2f" % (s2[r, 3] - s2[r, 2]).mean())
print("   state, LDS hand-over                     %.2f" % (sv[r, 2] - s2[r, 3]).mean())
print("   publish                                  %.2f" % (sv[r, 3] - sv[r, 2]).mean())
