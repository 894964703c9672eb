"""Same-box A/B of library builds on c2 / c1 / c4 (frames resident, counters off): alternates the libraries three times.
    python profiles/ab_probe.py libA libB ..."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
name, params, K = sys.argv[1], sys.argv[2], int(sys.argv[3])
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
p = syn.PARAMS[params]
f = sage.Frame(w["map"], w["scan"])
sage.set_counting(False)
run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
for _ in range(3): pose, st = run()
t = time.perf_counter()
for _ in range(K): pose, st = run()
dt = (time.perf_counter() - t) / K
print("%-30s %s %s: %8.3f ms/frame %4d it %6.2f us/it" % (os.path.basename(os.environ.get("SAGEICP_VARIANT_LIB", "product")), name, params, 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations)), flush=True)
'''
for wl in [tuple(x.split(":")) for x in os.environ.get("AB_WORKLOADS", "c2:cold:20 c1:cold:100 c4:steady:6").split()]:
    for rep in range(3):
        for lib in sys.argv[1:]:
            env = dict(os.environ)
            if lib != "product":
                env["SAGEICP_VARIANT_LIB"] = lib
            subprocess.run([sys.executable, "-c", CHILD, *wl], env=env, timeout=900)
