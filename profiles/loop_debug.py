"""First-aid for the one-launch loop: per shape, iteration count, correspondences per iteration and pose of
k_loop against the launch-per-iteration loop at the same lanes per query.
    python profiles/loop_debug.py [workload] [scale] [params]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402
from sage_icp_amd import synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
p = syn.PARAMS[sys.argv[3] if len(sys.argv) > 3 else "cold"]
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0), scale=scale)
KNOBS = ("SAGEICP_LOOP", "SAGEICP_LW", "SAGEICP_LOOP_WAVES", "SAGEICP_LOOP_GPW", "SAGEICP_FILTER", "SAGEICP_LOOP_CONTIGUOUS")
os.environ["SAGEICP_LOOP_DEBUG"] = "1"


def run(**env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    return sage.register_frame(w["scan"], w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)


print("%s x %.2f: %d queries" % (name, scale, len(w["scan"])))
for lw in (1, 2, 3, 4):
    for filt in (0, 1):
        a, sa = run(SAGEICP_LOOP=0, SAGEICP_LW=lw, SAGEICP_FILTER=filt)
        for extra in ({}, {"SAGEICP_LOOP_WAVES": 7, "SAGEICP_LOOP_GPW": 3}, {"SAGEICP_LOOP_WAVES": 1}):
            try:
                b, sb = run(SAGEICP_LOOP=2, SAGEICP_LW=lw, SAGEICP_FILTER=filt, **extra)
            except Exception as e:  # noqa: BLE001
                print("LW=%d filt=%d %s: %s" % (lw, filt, extra, e))
                continue
            ok = np.array_equal(a, b)
            print("LW=%d filt=%d %-50s one_launch=%d it %d/%d  %s" % (lw, filt, extra, sb.single_launch, sb.iterations, sa.iterations,
                                                                     "== " if ok else "DIFFERS max %.3e" % np.abs(a - b).max()), flush=True)
            if not ok:
                print("    n_corr k_icp :", list(sa.n_corr_hist)[:8])
                print("    n_corr k_loop:", list(sb.n_corr_hist)[:8])
                print("    cand %d / %d   pairs %d / %d" % (sa.sum_candidates, sb.sum_candidates, sa.pairs_evaluated, sb.pairs_evaluated))
