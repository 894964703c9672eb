"""both loops stopped after k iterations (SAGEICP_MAX_ITER): the state each left — which sums differ?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

name, scale = "c2", 0.05
p = syn.PARAMS["cold"]
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0), scale=scale)
NAMES = ["W", "Wsx", "Wsy", "Wsz", "Wxx", "Wxy", "Wxz", "Wyy", "Wyz", "Wzz", "Wrx", "Wry", "Wrz", "Wcx", "Wcy", "Wcz", "count"]


def state():
    out = np.zeros(36)
    sage.lib().sageicp_debug_last_state(C.c_void_p(w["map"]._h), out.ctypes.data_as(C.c_void_p))
    return out


for lw in (2, 3):
    for k in (1, 2):
        res = {}
        for loop in (0, 2):
            os.environ.update(SAGEICP_LOOP=str(loop), SAGEICP_LW=str(lw), SAGEICP_MAX_ITER=str(k))
            b, sb = sage.register_frame(w["scan"], w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            res[loop] = (b, sb, state())
        a, b = res[0][2], res[2][2]
        print("LW=%d after %d iteration(s): one_launch=%d; iterations %d / %d" % (lw, k, res[2][1].single_launch, a[34], b[34]))
        print("   T     k_icp ", np.array2string(a[:7], precision=6))
        print("   T     k_loop", np.array2string(b[:7], precision=6))
        print("   T_icp k_icp ", np.array2string(a[7:14], precision=6))
        print("   T_icp k_loop", np.array2string(b[7:14], precision=6))
        for i, nm in enumerate(NAMES):
            print("   %-6s %22.12e %22.12e %s" % (nm, a[14 + i], b[14 + i], "" if a[14 + i] == b[14 + i] else "  <-- differs"))
