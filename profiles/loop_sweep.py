"""Sweep of the one-launch loop's shape (k_loop: lanes per query, waves per workgroup, XCD mapping) against the
launch-per-iteration loop on one workload, frame resident.
    python profiles/loop_sweep.py [workload c2] [params cold] [divisor 1] [K 5]
Environment knobs are re-read by the library at every call, so one process serves every setting."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

if os.environ.get("LOOP_LIB"):
    sage.LIB_PATH = os.path.abspath(os.environ["LOOP_LIB"])
from sage_icp_amd import synthetic as syn  # noqa: E402

KNOBS = ("SAGEICP_LOOP", "SAGEICP_LW", "SAGEICP_LOOP_WAVES", "SAGEICP_LOOP_GPW", "SAGEICP_FILTER", "SAGEICP_LOOP_CONTIGUOUS")


def timed(frame, m, p, K, **env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    os.environ["SAGEICP_LOOP_COOLDOWN"] = "0"      # (a shape that times out must not keep the next one from being tried)
    run = lambda: sage.register_frame(frame, m, sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],  # noqa: E731
                                      return_stats=True)
    for _ in range(2):
        pose, st = run()
    t = time.perf_counter()
    for _ in range(K):
        pose, st = run()
    dt = (time.perf_counter() - t) / K
    return pose, st, dt


LWS = tuple(int(x) for x in os.environ.get("SWEEP_LW", "2,3,1,4").split(","))
CONTIG = tuple(int(x) for x in os.environ.get("SWEEP_CONTIG", "0,1").split(","))
NWS = tuple(int(x) for x in os.environ.get("SWEEP_NW", "0,4,8,7").split(","))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    params = sys.argv[2] if len(sys.argv) > 2 else "cold"
    div = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
    p = syn.PARAMS[params]
    n = len(w["scan"]) // div
    f = sage.Frame(w["map"], w["scan"][:n])
    print("%s %s, %d queries" % (name, params, n), flush=True)
    refs = {}
    for lw in (1, 2, 3, 4):
        pose, st, dt = timed(f, w["map"], p, K, SAGEICP_LOOP=0, SAGEICP_LW=lw)
        refs[lw] = pose
        print("  launch per iteration LW=%d %34s %8.3f ms/frame %4d it %6.2f us/it" % (lw, "", 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations)), flush=True)
    pose, st, dt = timed(f, w["map"], p, K, SAGEICP_LOOP=0)
    print("  launch per iteration (default: %d lanes) %22s %8.3f ms/frame %4d it %6.2f us/it" % (st.lanes_per_query, "", 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations)), flush=True)
    pose, st, dt = timed(f, w["map"], p, K)
    print("  library default: %s, %d lanes %31s %8.3f ms/frame %4d it %6.2f us/it" % ("one launch" if st.single_launch else "launch per iteration", st.lanes_per_query, "", 1e3 * dt, st.iterations, 1e6 * dt / max(1, st.iterations)), flush=True)
    for lw in LWS:
        for nw in NWS:
            for contig in CONTIG:
                for gpw in ((0,) if nw else (0,)):
                    env = dict(SAGEICP_LOOP=2, SAGEICP_LW=lw, SAGEICP_LOOP_CONTIGUOUS=contig)
                    if nw:
                        env["SAGEICP_LOOP_WAVES"] = nw
                    try:
                        pose, st, dt = timed(f, w["map"], p, K, **env)
                    except Exception as e:  # noqa: BLE001
                        print("  LW=%d nw=%d contig=%d: %s" % (lw, nw, contig, e), flush=True)
                        continue
                    same = "pose == launch-per-iteration" if np.array_equal(pose, refs[lw]) else "POSE DIFFERS (%.2e)" % np.abs(pose - refs[lw]).max()
                    print("  %-12s LW=%d nw=%s contig=%d %-22s %8.3f ms/frame %4d it %6.2f us/it  %s"
                          % ("one launch" if st.single_launch else "(fell back)", lw, nw or "auto", contig, "", 1e3 * dt, st.iterations,
                             1e6 * dt / max(1, st.iterations), same), flush=True)


if __name__ == "__main__":
    main()
