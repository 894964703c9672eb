"""Register a 1/div share of the c2 frame a few times (target of rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
div = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
n = len(w["scan"]) // div
f = sage.Frame(w["map"], w["scan"][:n])
for _ in range(reps):
    t = time.perf_counter()
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    dt = time.perf_counter() - t
print("%d queries: %.3f ms, %d iterations, %.1f us/iteration, lanes/query %d, pairs/cand %.3f" % (
    n, 1e3 * dt, st.iterations, 1e6 * dt / st.iterations, st.lanes_per_query, st.pairs_evaluated / max(1, st.sum_candidates)))
