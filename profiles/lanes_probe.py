"""Lanes per query of k_icp on frames of 50k .. 120k (c2 subsets) and 100k .. 500k (c4 subsets) queries: where
does two-lanes-per-query overtake four?  (the launch-per-iteration loop: SAGEICP_LOOP=0)
    python profiles/lanes_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
os.environ["SAGEICP_LOOP"] = "0"
for name, voxel, sizes in (("c2", 1.0, (50000, 65000, 80000, 100000, 120000)), ("c4", 1.0, (120000, 200000, 300000, 400000, 500000))):
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(voxel, 100.0))
    for prm in ("cold", "steady"):
        p = syn.PARAMS[prm]
        for n in sizes:
            f = sage.Frame(w["map"], w["scan"][:n])
            row = []
            for lw in (1, 2):
                os.environ["SAGEICP_LW"] = str(lw)
                run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
                for _ in range(2): run()
                K = 4 if name == "c2" else 2
                t = time.perf_counter()
                for _ in range(K): pose, st = run()
                dt = (time.perf_counter() - t) / K
                row.append("lw=%d %7.3f ms %5.1f us/it" % (lw, 1e3 * dt, 1e6 * dt / st.iterations))
            print("%s %s n=%6d %3d it | %s" % (name, prm, n, st.iterations, " | ".join(row)), flush=True)
