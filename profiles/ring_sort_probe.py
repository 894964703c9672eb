"""kSortFrameFrom on the SECOND scene family (ring geometry): frames of a few sizes searched as they came against sorted.
    python profiles/ring_sort_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

sage.set_counting(False)
w = syn.make_ring_workload(lambda: sage.VoxelHashMap(1.0, 100.0), n_map_scans=40)
guess = w["T_gt"].copy()
guess[4] -= 0.5
rng = np.random.default_rng(3)
scan = w["scan"][rng.permutation(len(w["scan"]))]          # (a pipeline hands over hash-map order: spatially random)
p = syn.PARAMS["cold"]
for n in (5000, 10000, 15000, 20000, 26000, 40000, 60000, 80000, len(scan)):
    f = sage.Frame(w["map"], scan[:n])
    res = {}
    for label, val in (("sorted", "0"), ("as it came", "1000000")):
        os.environ["SAGEICP_SORT_FROM"] = val
        run = lambda: sage.register_frame(f, w["map"], guess, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)  # noqa: E731
        for _ in range(3):
            pose, st = run()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(20):
                pose, st = run()
            best = min(best, (time.perf_counter() - t) / 20)
        res[label] = (best, st.iterations, st.lanes_per_query, pose)
    a, b = res["sorted"], res["as it came"]
    print("ring %6d points: sorted %.3f ms (%d it, %d lanes)  as it came %.3f ms (%d it)  %+.1f %%  pose delta %.1e"
          % (n, 1e3 * a[0], a[1], a[2], 1e3 * b[0], b[1], 100 * (b[0] / a[0] - 1), np.abs(a[3] - b[3]).max()), flush=True)
