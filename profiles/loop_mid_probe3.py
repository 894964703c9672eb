"""Mid-size frames after the lane rules of late round 4: default choice (SAGEICP_LOOP=1) against the launch-per-iteration
loop (=0), and the compact scan on / off inside the one-launch loop.    python profiles/loop_mid_probe3.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
for prm in ("cold", "steady"):
    p = syn.PARAMS[prm]
    for n in (24000, 32000, 40000, 50000, 60000, 70000, 90000, 110000, 120000):
        f = sage.Frame(w["map"], w["scan"][:n])
        row = []
        for mode, filt in ((0, None), (1, None), (1, 0), (1, 1)):
            os.environ["SAGEICP_LOOP"] = str(mode)
            os.environ.pop("SAGEICP_FILTER", None)
            if filt is not None:
                os.environ["SAGEICP_FILTER"] = str(filt)
            run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            for _ in range(3): run()
            t = time.perf_counter()
            for _ in range(6): pose, st = run()
            dt = (time.perf_counter() - t) / 6
            row.append("LOOP=%d%s %5.1f us/it [%s, %d lanes%s]" % (mode, "" if filt is None else " FILTER=%d" % filt, 1e6 * dt / st.iterations,
                                                               "one launch" if st.single_launch else "per iteration", st.lanes_per_query, ", compact" if st.compact_scan else ""))
        print("c2 %s n=%6d %3d it | %s" % (prm, n, st.iterations, " | ".join(row)), flush=True)
