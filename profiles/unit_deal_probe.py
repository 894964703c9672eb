"""Would dealing the sorted frame's units round-robin to the workgroups (instead of a contiguous run of units per workgroup)
combine the order's locality with the balance of a random order?  Emulated from outside: the frame sorted here (Morton key
of the voxel under the initial guess), its units of U consecutive queries permuted with a stride, handed over with the
library's own sort off.  Both scene families.
    python profiles/unit_deal_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

sage.set_counting(False)


def morton_order(scan, guess, vs):
    import oracle
    p = oracle.transform_points(guess, scan)[:, :3]
    c = (np.trunc(p / vs).astype(np.int64) + 512) & 1023

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    return np.argsort(key, kind="stable")


def timed(w, frame, guess, p, K):
    f = sage.Frame(w["map"], np.ascontiguousarray(frame))
    run = lambda: sage.register_frame(f, w["map"], guess, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)  # noqa: E731
    for _ in range(3):
        pose, st = run()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(K):
            pose, st = run()
        best = min(best, (time.perf_counter() - t) / K)
    return best, st


def probe(name, w, scan, guess, p, vs, K):
    rng = np.random.default_rng(5)
    scan = scan[rng.permutation(len(scan))]
    os.environ["SAGEICP_SORT_FROM"] = "0"
    t_sorted, st = timed(w, scan, guess, p, K)
    U = 64 // st.lanes_per_query
    os.environ["SAGEICP_SORT_FROM"] = "100000000"
    t_random, _ = timed(w, scan, guess, p, K)
    srt = scan[morton_order(scan, guess, vs)]
    t_mine, _ = timed(w, srt, guess, p, K)                       # (check: my sort handed over as it is = the library's)
    nu = len(srt) // U
    out = ["%-22s %6d pts, %2d lanes: library's sort %.3f ms | as it came %+.1f %% | sorted here %+.1f %%"
           % (name, len(scan), st.lanes_per_query, 1e3 * t_sorted, 100 * (t_random / t_sorted - 1), 100 * (t_mine / t_sorted - 1))]
    for S in (8, 64, 416, 1664):
        if S >= nu:
            continue
        order = np.concatenate([np.arange(r, nu, S) for r in range(S)])
        idx = (order[:, None] * U + np.arange(U)[None, :]).ravel()
        idx = np.concatenate([idx, np.arange(nu * U, len(srt))])
        t, _ = timed(w, srt[idx], guess, p, K)
        out.append("units dealt with stride %4d %+.1f %%" % (S, 100 * (t / t_sorted - 1)))
    print(" | ".join(out), flush=True)


wr = syn.make_ring_workload(lambda: sage.VoxelHashMap(1.0, 100.0), n_map_scans=40)
g = wr["T_gt"].copy()
g[4] -= 0.5
for n in (26000, 60000, len(wr["scan"])):
    probe("ring", wr, wr["scan"][:n] if n < len(wr["scan"]) else wr["scan"], g, syn.PARAMS["cold"], 1.0, 12)
del wr
w2 = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
for div in (4, 2, 1):
    probe("street c2/%d" % div, w2, w2["scan"][: len(w2["scan"]) // div], sage.IDENTITY, syn.PARAMS["cold"], 1.0, 8)
