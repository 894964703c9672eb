"""Per-iteration time of one rank's share of the c2 (or, `shard_probe.py c4`, the c4) frame (the
first 1/N of the queries against the full map), without any exchange: the compute side of the
strong-scaling curve."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = syn.make_workload(wl, lambda: sage.VoxelHashMap(syn.WORKLOADS[wl]["voxel"], 100.0))
p = syn.PARAMS[sys.argv[2] if len(sys.argv) > 2 else "cold"]
for N in (1, 2, 4, 8):
    n = len(w["scan"]) // N
    f = sage.Frame(w["map"], w["scan"][:n])
    for loop in (0, 1):           # the launch-per-iteration loop | the library's choice (the one-launch loop where the shard fits)
        os.environ["SAGEICP_LOOP"] = str(loop)
        for _ in range(3): sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
        t = time.perf_counter(); K = 20 if wl == "c2" else 6
        for _ in range(K): pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        dt = (time.perf_counter() - t) / K
        print("N=%d  %6d queries  %-22s %d lanes  %.3f ms/frame  %d iterations  %.1f us/iteration"
              % (N, n, "one launch" if st.single_launch else "launch per iteration", st.lanes_per_query, 1e3 * dt, st.iterations, 1e6 * dt / st.iterations), flush=True)
