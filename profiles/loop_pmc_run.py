"""one k_loop registration of a c2 shard under rocprofv3 --pmc (instructions per wave and iteration)
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES ... -- python profiles/loop_pmc_run.py [div] [lw]"""
import os, sys
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
div = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:
    os.environ["SAGEICP_LW"] = sys.argv[2]
os.environ["SAGEICP_LOOP"] = "2"
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
n = len(w["scan"]) // div
f = sage.Frame(w["map"], w["scan"][:n])
for _ in range(3):
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
print("queries", n, "lanes", st.lanes_per_query, "iterations", st.iterations, "one launch", st.single_launch)
