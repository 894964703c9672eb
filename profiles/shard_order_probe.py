"""Strong-scaling shards of the c2 frame: blocks of the scan in arrival order vs blocks of the
Morton-sorted scan (spatially compact).  Per block: microseconds per ICP iteration on one GPU."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
scan = w["scan"]
def morton(pts, vs=1.0):
    k = np.trunc(pts[:, :3] / vs).astype(np.int64); k -= k.min(0)
    def part(x):
        x = x & 0x3FF; x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249; return x
    return np.argsort(part(k[:, 0]) | (part(k[:, 1]) << 1) | (part(k[:, 2]) << 2), kind="stable")
for N in (2, 4, 8):
    for name, s in (("arrival", scan), ("morton", scan[morton(scan)])):
        per = []
        for b in range(N):
            lo, hi = b * len(s) // N, (b + 1) * len(s) // N
            f = sage.Frame(w["map"], s[lo:hi])
            for _ in range(2): sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
            t = time.perf_counter(); K = 8
            for _ in range(K): pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
            per.append(1e6 * (time.perf_counter() - t) / K / st.iterations)
        print("N=%d %-8s us/iteration per block: %s  max %.1f mean %.1f" % (N, name, " ".join("%.1f" % x for x in per), max(per), sum(per) / N), flush=True)
