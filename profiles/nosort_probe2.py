"""Where does the frame's sort start to pay?  Parts of the c2 scan (random order) sorted / in the caller's order."""
import os, subprocess, sys
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(syn.WORKLOADS["c2"]["voxel"], 100.0))
p = syn.PARAMS["cold"]
sage.set_counting(False)
for n in (5000, 10000, 15000, 20000, 30000, 45000, 60000):
    f = sage.Frame(w["map"], w["scan"][:n])
    out = []
    for nosort in (0, 1):
        if nosort: os.environ["SAGEICP_NO_SORT"] = "1"
        else: os.environ.pop("SAGEICP_NO_SORT", None)
        run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        for _ in range(3): pose, st = run()
        t = time.perf_counter(); K = 30
        for _ in range(K): pose, st = run()
        out.append((1e3 * (time.perf_counter() - t) / K, st.iterations, st.lanes_per_query))
    print("%6d queries: sorted %.3f ms (%d it, %d lanes) | caller's order %.3f ms (%d it)  -> %+.1f %%" % (n, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], 100 * (out[1][0] / out[0][0] - 1)), flush=True)
'''
env = dict(os.environ)
env["SAGEICP_VARIANT_LIB"] = sys.argv[1]
subprocess.run([sys.executable, "-c", CHILD], env=env, timeout=1500)
