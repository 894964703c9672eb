"""Soak of the one-launch loop: thousands of registrations back to back — every one must run in one launch
(no wait inside it timing out) and give the same pose to the bit.
    python profiles/loop_soak.py [seconds per workload, default 60]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cases = []
w1 = syn.make_workload("c1", lambda: sage.VoxelHashMap(0.8, 100.0))
cases.append(("c1 cold", w1, w1["scan"], syn.PARAMS["cold"]))
w2 = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
for div, prm in ((8, "cold"), (5, "steady"), (2, "cold"), (1, "cold")):       # (1: the headline frame, two passes per workgroup)
    cases.append(("c2/%d %s" % (div, prm), w2, w2["scan"][: len(w2["scan"]) // div], syn.PARAMS[prm]))
if os.environ.get("SOAK_CHAINED", "1") != "0":
    # (round 6) a frame beyond the LDS: the launches chained beside the resident solving wave — every registration must stay chained
    w5 = syn.make_workload("c5", lambda: sage.VoxelHashMap(syn.WORKLOADS["c5"]["voxel"], 100.0))
    cases.append(("c5 dense", w5, w5["scan"], syn.PARAMS["dense"]))
for name, w, scan, p in cases:
    f = sage.Frame(w["map"], scan)
    ref, st0 = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    n = fallbacks = differs = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        n += 1
        fallbacks += 0 if st.single_launch else 1
        differs += 0 if np.array_equal(pose, ref) else 1
    dt = time.perf_counter() - t0
    ls = w["map"].loop_status()
    if not st0.single_launch:
        # chained form: what counts is that no call fell back to the form with k_fin (timeouts) — the handle's own counters say
        fallbacks = int(ls.timeouts) + (0 if ls.calls_chained >= n else 1)
    print("%-14s %6d queries, %2d lanes/query: %6d registrations in %.0f s (%.3f ms each, %d iterations), launches that gave up: %d, poses that differ: %d%s"
          % (name, len(scan), st0.lanes_per_query, n, dt, 1e3 * dt / n, st0.iterations, fallbacks, differs,
             "" if st0.single_launch else "  (chained: %d calls)" % ls.calls_chained), flush=True)
