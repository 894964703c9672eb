"""How many workgroups of k_loop are resident at once?  The c2 frame in one launch with the grid capped at
N workgroups (SAGEICP_LOOP_MAX_WGS): a grid that is not resident as a whole times out (50 ms) and falls back."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
f = sage.Frame(w["map"], w["scan"])
os.environ.update(SAGEICP_LOOP="2", SAGEICP_LOOP_COOLDOWN="0", SAGEICP_LOOP_DEBUG="1")
for nw in (4, 8):
    for n in ((1504, 1536, 1568, 1600, 1632, 1664, 1696, 1728, 1760) if nw == 4 else (640, 672, 704, 736, 768)):
        os.environ.update(SAGEICP_LOOP_WAVES=str(nw), SAGEICP_LOOP_MAX_WGS=str(n))
        for _ in range(2):
            t = time.perf_counter()
            pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            dt = time.perf_counter() - t
        print("nw=%d max %4d workgroups: %s  %.3f ms" % (nw, n, "one launch" if st.single_launch else "TIMED OUT", 1e3 * dt), flush=True)
