import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
name = sys.argv[1]
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
f = sage.Frame(w["map"], w["scan"]); p = syn.PARAMS["cold"]
for _ in range(3): sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
t = time.perf_counter(); N = 20
for _ in range(N): pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
dt = (time.perf_counter() - t) / N
print(name, os.environ.get("SAGEICP_CHUNKED", "0"), os.environ.get("SAGEICP_DEPTH", "-"), "ms/frame %.3f  iters %d resorts %d  us/iter %.2f" % (1e3 * dt, st.iterations, st.resorts, 1e6 * dt / st.iterations))
