"""Sixteen against eight lanes per query on small frames against dense voxels (c2 subsets), both loops.
    python profiles/lanes_probe4.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
for prm in ("cold", "steady"):
    p = syn.PARAMS[prm]
    for n in (2000, 5000, 8000, 10000, 12000, 15000):
        f = sage.Frame(w["map"], w["scan"][:n])
        row = []
        for mode, lw in ((0, 3), (0, 4), (2, 3), (2, 4), (1, None)):
            os.environ["SAGEICP_LOOP"] = str(mode)
            os.environ.pop("SAGEICP_LW", None)
            if lw is not None: os.environ["SAGEICP_LW"] = str(lw)
            run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            for _ in range(3): run()
            t = time.perf_counter()
            for _ in range(8): pose, st = run()
            dt = (time.perf_counter() - t) / 8
            row.append("%s%s %5.1f" % ("one launch " if st.single_launch else "per it ", "default(%d)" % st.lanes_per_query if lw is None else "%d lanes" % st.lanes_per_query, 1e6 * dt / st.iterations))
        print("c2 %s n=%6d %3d it | us/it: %s" % (prm, n, st.iterations, " | ".join(row)), flush=True)
