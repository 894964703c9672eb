"""Eight against four lanes per query in the launch-per-iteration loop (8 lanes stride in flat order since late round 4)
on frames of 40k .. 120k queries: where does the switch to four belong now?    python profiles/lanes_probe2.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
os.environ["SAGEICP_LOOP"] = "0"
for name, sizes in (("c2", (40000, 50000, 60000, 70000, 80000, 100000, 120000)), ("c4", (60000, 120000))):
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(1.0, 100.0))
    for prm in ("cold", "steady"):
        p = syn.PARAMS[prm]
        for n in sizes:
            f = sage.Frame(w["map"], w["scan"][:n])
            row = []
            for lw in (2, 3):
                for filt in (0, 1):
                    os.environ["SAGEICP_LW"] = str(lw)
                    os.environ["SAGEICP_FILTER"] = str(filt)
                    run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
                    for _ in range(2): run()
                    t = time.perf_counter()
                    for _ in range(4): pose, st = run()
                    dt = (time.perf_counter() - t) / 4
                    row.append("lw=%d%s %5.1f" % (lw, "c" if filt else "f", 1e6 * dt / st.iterations))
            print("%s %s n=%6d %3d it | us/it: %s" % (name, prm, n, st.iterations, " | ".join(row)), flush=True)
