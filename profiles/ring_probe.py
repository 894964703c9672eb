"""The library's break-points (lanes per query, compact-scan filter, flat order) on the SECOND scene family: scans of
64-beam ring geometry against a map built from such scans (synthetic.make_ring_workload).  For each frame size: the
library's own choice against every forced alternative; poses compared bit for bit.
    python profiles/ring_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

sage.set_counting(False)
w = syn.make_ring_workload(lambda: sage.VoxelHashMap(1.0, 100.0), n_map_scans=40)
print("ring family: map %d points in %d voxels (%.1f per voxel), scan %d points" % (w["map"].size(), w["map"].num_voxels(),
      w["map"].size() / w["map"].num_voxels(), len(w["scan"])))
guess = w["T_gt"].copy()
guess[4] -= 0.5
KNOBS = ("SAGEICP_LW", "SAGEICP_FILTER", "SAGEICP_FLAT", "SAGEICP_LOOP")


def timed(f, p, K, **env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    run = lambda: sage.register_frame(f, w["map"], guess, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)  # noqa: E731
    for _ in range(2):
        pose, st = run()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(K):
            pose, st = run()
        best = min(best, (time.perf_counter() - t) / K)
    return pose, st, best


for params in ("cold", "steady"):
    p = syn.PARAMS[params]
    for div in (1, 2, 4, 10):
        scan = np.ascontiguousarray(w["scan"][::div])
        f = sage.Frame(w["map"], scan)
        K = max(4, 40 // max(1, 4 // div))
        ref, st0, t0 = timed(f, p, K)
        print("%s, %d points: library's choice %d lanes, %s scan, %s: %.3f ms (%d it, %.2f us/it)" % (
            params, len(scan), st0.lanes_per_query, "compact" if st0.compact_scan else "full", "one launch" if st0.single_launch else "k_icp + k_fin",
            1e3 * t0, st0.iterations, 1e6 * t0 / st0.iterations), flush=True)
        for lw in (1, 2, 3, 4):
            pose, st, t = timed(f, p, K, SAGEICP_LW=lw)
            print("     %2d lanes: %+6.1f %%  %s%s" % (1 << lw, 100 * (t / t0 - 1), "one launch" if st.single_launch else "k_icp + k_fin",
                                                      "" if np.array_equal(pose, ref) else "  POSE DIFFERS"), flush=True)
        pose, st, t = timed(f, p, K, SAGEICP_FILTER=0 if st0.compact_scan else 1)
        print("     %s scan: %+6.1f %%%s" % ("full" if st0.compact_scan else "compact", 100 * (t / t0 - 1), "" if np.array_equal(pose, ref) else "  POSE DIFFERS"), flush=True)
        if st0.lanes_per_query <= 4:
            for fl in (0, 1):
                pose, st, t = timed(f, p, K, SAGEICP_FLAT=fl, SAGEICP_LOOP=0)
                print("     k_icp + k_fin, flat order %d: %+6.1f %%%s" % (fl, 100 * (t / t0 - 1), "" if np.array_equal(pose, ref) else "  POSE DIFFERS"), flush=True)
