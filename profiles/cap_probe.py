"""Timing probe: how much of an iteration is the heaviest queries' scans?  Variant builds (-DSAGE_CAP_PROBE=k) in which no
query looks at more than k voxels in its main scan (WRONG answers) and the loop never converges (200 iterations).
    python profiles/cap_probe.py lib [lib ...]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(syn.WORKLOADS["c2"]["voxel"], 100.0))
p = syn.PARAMS["cold"]
f = sage.Frame(w["map"], w["scan"])
sage.set_counting(False)
run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
for tri in ("0", "1"):
    os.environ["SAGEICP_LOOP_TRI"] = tri
    for _ in range(2): pose, st = run()
    t = time.perf_counter(); K = 4
    for _ in range(K): pose, st = run()
    dt = (time.perf_counter() - t) / K
    print("%-28s %d lanes: %8.3f ms/frame %4d it %6.2f us/it" % (os.path.basename(os.environ.get("SAGEICP_VARIANT_LIB", "product")), st.lanes_per_query, 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations)), flush=True)
'''
for lib in sys.argv[1:]:
    env = dict(os.environ)
    env["SAGEICP_VARIANT_LIB"] = lib
    env["SAGEICP_MAX_ITER"] = "200"
    subprocess.run([sys.executable, "-c", CHILD], env=env, timeout=600)
