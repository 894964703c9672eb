"""Lanes per query against SPARSE voxels (c5: 0.1 m voxels, ~2 points each; c1: 0.8 m, a few) after the flat-order scan:
launch-per-iteration loop with 2 / 4 / 8 lanes, and what the default (one-launch loop where it fits) does.
    python profiles/lanes_probe3.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
for name, voxel, prm, sizes in (("c5", 0.1, "dense", (12000, 25000, 50000, 100000, 200000)), ("c1", 0.8, "cold", (5000, 10000))):
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(voxel, 100.0))
    p = syn.PARAMS[prm]
    for n in sizes:
        if n > len(w["scan"]): continue
        f = sage.Frame(w["map"], w["scan"][:n])
        row = []
        for mode, lw in ((0, 1), (0, 2), (0, 3), (1, None)):
            os.environ["SAGEICP_LOOP"] = str(mode)
            os.environ.pop("SAGEICP_LW", None)
            if lw is not None: os.environ["SAGEICP_LW"] = str(lw)
            run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            for _ in range(3): run()
            t = time.perf_counter()
            for _ in range(6): pose, st = run()
            dt = (time.perf_counter() - t) / 6
            row.append("%s %5.1f us/it [%s, %d lanes]" % ("default" if lw is None else "lw=%d" % lw, 1e6 * dt / st.iterations, "one launch" if st.single_launch else "per iteration", st.lanes_per_query))
        print("%s n=%6d %3d it | %s" % (name, n, st.iterations, " | ".join(row)), flush=True)
