"""Experiment (round 4, not kept): another lanes-per-query variant of k_icp from some iteration on (the late
iterations of a registration scan little).  Needs the two-line hack in capi.hip::run_icp that reads SAGEICP_LW_LATE /
SAGEICP_LW_SWITCH (removed again); result: profiles/r04/lw_late.txt — four lanes per query throughout stays best on c2
(7.57 ms; two lanes from iteration 80 on: 7.93), two on c4."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
for wl, prm in (("c2", "cold"), ("c2", "steady"), ("c4", "steady")):
    w = syn.make_workload(wl, lambda: sage.VoxelHashMap(syn.WORKLOADS[wl]["voxel"], 100.0))
    p = syn.PARAMS[prm]
    f = sage.Frame(w["map"], w["scan"])
    print(wl, prm, flush=True)
    base = 2 if wl == "c2" else 1
    for late, sw in ((base, 1 << 30), (1, 0), (1, 20), (1, 40), (1, 80), (3, 40), (0, 40), (0, 80)) if wl == "c2" else ((base, 1 << 30), (0, 20), (0, 60), (2, 0), (2, 40)):
        os.environ["SAGEICP_LW_LATE"] = str(late); os.environ["SAGEICP_LW_SWITCH"] = str(sw)
        run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        for _ in range(2): run()
        K = 6 if wl == "c2" else 3
        t = time.perf_counter()
        for _ in range(K): pose, st = run()
        dt = (time.perf_counter() - t) / K
        print("  late LW=%d from iteration %-10d %8.3f ms/frame %d it" % (late, sw, 1e3 * dt, st.iterations), flush=True)
