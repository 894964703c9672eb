"""Per-kernel time of one ICP iteration (HIP events, profiling level 2) for a workload."""
import os, sys
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
f = sage.Frame(w["map"], w["scan"]); p = syn.PARAMS["cold"]
sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
sage.set_profiling(2)
pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
n = st.nn_launches
print("%s: %d queries, %d iterations; per iteration us: k_group %.1f  k_nn %.1f  k_gn(+finish) %.1f  | wall/iter %.1f"
      % (name, st.n_queries, st.iterations, st.us_group / n, st.us_nn / n, st.us_gn / n, st.us_wall / st.iterations))
