"""A/B of k_skip against k_icp through bench-like timing: frames resident, K registrations each.
    python profiles/skip_ab.py [workload params]..."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

cases = [a.split(":") for a in sys.argv[1:]] or [["c2", "cold"], ["c2", "steady"], ["c5", "dense"], ["c4", "steady"]]
for wl, prm in cases:
    w = syn.make_workload(wl, lambda: sage.VoxelHashMap(syn.WORKLOADS[wl]["voxel"], 100.0))
    p = syn.PARAMS[prm]
    f = sage.Frame(w["map"], w["scan"])
    print("%s %s, %d queries" % (wl, prm, len(w["scan"])), flush=True)
    ref = None
    settings = [dict(SAGEICP_SKIP=0), dict(SAGEICP_SKIP=1), dict(SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=10),
                dict(SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=20), dict(SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=30),
                dict(SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=20, SAGEICP_FILTER=0),
                dict(SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=20, SAGEICP_LW=1), dict(SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=20, SAGEICP_LW=3)]
    for env in settings:
        for k in ("SAGEICP_SKIP", "SAGEICP_SKIP_MARGIN_MM", "SAGEICP_FILTER", "SAGEICP_LW"):
            os.environ.pop(k, None)
        os.environ["SAGEICP_LOOP"] = "0"
        for k, v in env.items():
            os.environ[k] = str(v)
        run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)  # noqa: E731
        for _ in range(2):
            pose, st = run()
        K = 6 if wl != "c4" else 3
        t = time.perf_counter()
        for _ in range(K):
            pose, st = run()
        dt = (time.perf_counter() - t) / K
        if ref is None:
            ref = pose
        print("  %-60s %8.3f ms/frame %4d it %6.1f us/it  searched %5.1f %%  pairs %5.1f %% of candidates  |dpose| %.1e"
              % (" ".join("%s=%s" % kv for kv in env.items()), 1e3 * dt, st.iterations, 1e6 * dt / st.iterations,
                 100.0 * st.queries_searched / (st.iterations * len(w["scan"])), 100.0 * st.pairs_evaluated / max(1, st.sum_candidates),
                 np.abs(pose - ref).max()), flush=True)
