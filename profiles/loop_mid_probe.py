import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
for prm in ("cold", "steady"):
    p = syn.PARAMS[prm]
    for div in (2, 3, 4):
        n = len(w["scan"]) // div
        f = sage.Frame(w["map"], w["scan"][:n])
        for mode in (0, 1):
            os.environ["SAGEICP_LOOP"] = str(mode)
            run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            for _ in range(3): run()
            t = time.perf_counter()
            for _ in range(10): pose, st = run()
            dt = (time.perf_counter() - t) / 10
            print("c2 %s %6d queries LOOP=%d: %.3f ms %d it %.1f us/it one_launch=%d lanes=%d" % (prm, n, mode, 1e3*dt, st.iterations, 1e6*dt/st.iterations, st.single_launch, st.lanes_per_query), flush=True)
