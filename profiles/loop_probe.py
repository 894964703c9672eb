"""A/B of the one-launch ICP loop (k_loop) against the launch-per-iteration loop (k_icp + k_fin): wall time
per RegisterFrame and per iteration, frame resident, for the frame sizes the one-launch loop is meant
for (shards of the c2 frame, c1, and — forced to two lanes per query — the whole c2 frame).
    python profiles/loop_probe.py [quick]
Environment knobs are re-read by the library at every call, so one process serves every setting."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

if os.environ.get("LOOP_LIB"):           # a variant build of the library (A/B runs)
    sage.LIB_PATH = os.path.abspath(os.environ["LOOP_LIB"])
from sage_icp_amd import synthetic as syn  # noqa: E402

KNOBS = ("SAGEICP_LOOP", "SAGEICP_LW", "SAGEICP_LOOP_WAVES", "SAGEICP_FILTER")


def timed(frame, m, p, K, **env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    run = lambda: sage.register_frame(frame, m, sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],  # noqa: E731
                                      return_stats=True)
    for _ in range(3):
        pose, st = run()
    t = time.perf_counter()
    for _ in range(K):
        pose, st = run()
    dt = (time.perf_counter() - t) / K
    return pose, st, dt


def line(tag, pose, st, dt, ref=None):
    same = "" if ref is None else ("  pose == two-launch loop" if np.array_equal(pose, ref) else "  POSE DIFFERS")
    print("  %-44s %8.3f ms/frame %4d it %6.1f us/it  [%s, %d lanes/query%s]%s"
          % (tag, 1e3 * dt, st.iterations, 1e6 * dt / max(st.iterations, 1),
             "one launch" if st.single_launch else "launch per iteration", st.lanes_per_query,
             ", compact scan" if st.compact_scan else "", same), flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
    for params in ("cold", "steady"):
        p = syn.PARAMS[params]
        for div in (8, 4, 2, 1):
            n = len(w["scan"]) // div
            f = sage.Frame(w["map"], w["scan"][:n])
            K = 10 if div > 1 else 5
            print("c2 %s, %d queries (1/%d of the frame)" % (params, n, div))
            ref, st, dt = timed(f, w["map"], p, K, SAGEICP_LOOP=0)
            line("launch per iteration (default lanes)", ref, st, dt)
            lw0 = st.lanes_per_query.bit_length() - 1
            for lw in sorted({max(1, lw0 - 1), max(lw0, 1), min(lw0 + 1, 4)}):
                for nw in ((4, 8) if not quick else (4,)):
                    pose, st, dt = timed(f, w["map"], p, K, SAGEICP_LOOP=2, SAGEICP_LW=lw, SAGEICP_LOOP_WAVES=nw)
                    line("one launch, LW=%d, >=%d waves per workgroup" % (lw, nw), pose, st, dt, ref)
            if quick and div == 4:
                break
    w1 = syn.make_workload("c1", lambda: sage.VoxelHashMap(0.8, 100.0))
    p = syn.PARAMS["cold"]
    f = sage.Frame(w1["map"], w1["scan"])
    print("c1 cold, %d queries" % len(w1["scan"]))
    ref, st, dt = timed(f, w1["map"], p, 20, SAGEICP_LOOP=0)
    line("launch per iteration (default lanes)", ref, st, dt)
    for lw in (2, 3, 4):
        for nw in (1, 2, 4, 8):
            pose, st, dt = timed(f, w1["map"], p, 20, SAGEICP_LOOP=2, SAGEICP_LW=lw, SAGEICP_LOOP_WAVES=nw)
            line("one launch, LW=%d, >=%d waves per workgroup" % (lw, nw), pose, st, dt, ref)


if __name__ == "__main__":
    main()
