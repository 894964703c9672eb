"""Does the grid of the one-launch loop get bigger when the solving wave has a CU of its own (SAGEICP_SOLVER_CU, probe)?
    python profiles/solver_cu_probe.py     (c2 cold; one process per setting: the streams are made once per handle)"""
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
run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
for cap in (1664, 1696, 1728, 1760, 1792):
    os.environ["SAGEICP_LOOP_CAP_WGS"] = str(cap)
    os.environ["SAGEICP_LOOP_COOLDOWN"] = "0"
    for _ in range(2): pose, st = run()
    t = time.perf_counter(); K = 6
    for _ in range(K): pose, st = run()
    dt = (time.perf_counter() - t) / K
    print("solver CU %s  cap %d: %8.3f ms/frame %4d it %6.2f us/it  one launch %s" % (os.environ.get("SAGEICP_SOLVER_CU", "-"), cap, 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations), st.single_launch), flush=True)
'''
for solver_cu in (None, "1"):
    env = dict(os.environ)
    if solver_cu:
        env["SAGEICP_SOLVER_CU"] = solver_cu
    subprocess.run([sys.executable, "-c", CHILD], env=env, timeout=900)
