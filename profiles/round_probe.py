"""Does the c2 frame pay for not fitting ONE round of resident waves?  k_icp at four lanes per query puts
16 queries in a wave; 7 waves per SIMD x 1,024 SIMDs hold 7,168 waves = 114,688 queries.  Time per
iteration of the first n queries of the c2 frame, n around that edge (and with SAGEICP_LW forced)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
os.environ["SAGEICP_LOOP"] = "0"
for prm in ("cold", "steady"):
    p = syn.PARAMS[prm]
    for n in (100000, 108000, 112000, 114000, 114688, 115200, 116000, 118000, 120000):
        f = sage.Frame(w["map"], w["scan"][:n])
        run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        for _ in range(2): run()
        t = time.perf_counter()
        for _ in range(5): pose, st = run()
        dt = (time.perf_counter() - t) / 5
        print("%s n=%6d (%5d waves)  %7.3f ms/frame %3d it  %5.1f us/it  %.2f ns per query and iteration"
              % (prm, n, (n + 15) // 16, 1e3 * dt, st.iterations, 1e6 * dt / st.iterations, 1e9 * dt / st.iterations / n), flush=True)
