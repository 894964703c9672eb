"""Where a wave's pass through k_loop goes (cycles per phase, mean over all waves and iterations), from a build with
-DSAGE_LOOP_TIMING (sage-icp_amd/_probe/libsageicp_looptiming.so).
    python profiles/loop_phases.py [divisor of the frame, default 1] [cold|steady] [workload, default c2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

sage.LIB_PATH = os.environ.get("LOOP_LIB", os.path.join(os.path.dirname(sage.LIB_PATH), "_probe", "libsageicp_looptiming.so"))
from sage_icp_amd import synthetic as syn  # noqa: E402

div = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = syn.PARAMS[sys.argv[2] if len(sys.argv) > 2 else "cold"]
name = sys.argv[3] if len(sys.argv) > 3 else "c2"
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
n = len(w["scan"]) // div
f = sage.Frame(w["map"], w["scan"][:n])
sage.set_counting(False)
for _ in range(3):
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
assert st.single_launch == 1
ph = np.zeros(16, dtype=np.uint64)
sage.lib().sageicp_debug_loop_phases(ph.ctypes.data_as(C.c_void_p), 1)       # (read and reset)
pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
sage.lib().sageicp_debug_loop_phases(ph.ctypes.data_as(C.c_void_p), 1)
k = float(ph[10]) or 1.0
names = ["pose, query, home voxel", "row rebuild (stale), face gaps", "seed, home-only scan (unseeded)", "bounds -> need mask",
         "scan", "argmin", "answer's record (if changed)", "pair terms, wave reduction, ticket"]
ghz = 2.1
print("%s %s, %d queries, %d lanes/query, %d iterations; mean over %d wave-passes, us at %.1f GHz (s_memtime cycles):"
      % (name, sys.argv[2] if len(sys.argv) > 2 else "cold", n, st.lanes_per_query, st.iterations, int(k), ghz))
for i, nm in enumerate(names):
    print("   %-38s %6.2f" % (nm, ph[i] / k / ghz / 1e3))
print("   %-38s %6.2f   (sum of the above)" % ("body", ph[:8].sum() / k / ghz / 1e3))
print("   %-38s %6.2f   (last wave of a workgroup only)" % ("closing the workgroup", ph[9] / k / ghz / 1e3))
print("   %-38s %6.2f   (barrier: the workgroup's poller has the next pose)" % ("waiting", ph[8] / k / ghz / 1e3))
