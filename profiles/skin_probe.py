"""Timing probe: c2 cold through variant builds in which a share of the seeded queries skips its scan (wrong answers,
the cost of an iteration only).   python profiles/skin_probe.py lib [lib ...]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
name, params = sys.argv[1], sys.argv[2]
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
p = syn.PARAMS[params]
f = sage.Frame(w["map"], w["scan"])
run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
for _ in range(2): pose, st = run()
t = time.perf_counter()
K = 4
for _ in range(K): pose, st = run()
dt = (time.perf_counter() - t) / K
print("%-44s %s %s: %8.3f ms/frame %4d it %6.2f us/it  one launch %s" % (os.path.basename(os.environ.get("SAGEICP_VARIANT_LIB", "product")), name, params, 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations), st.single_launch), flush=True)
'''
for wl in (("c2", "cold"), ("c2", "steady"), ("c1", "cold")):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "product":
            env["SAGEICP_VARIANT_LIB"] = lib
        subprocess.run([sys.executable, "-c", CHILD, wl[0], wl[1]], env=env, timeout=600)
