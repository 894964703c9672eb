"""Same-process, interleaved A/B of environment knobs (the library re-reads them at every call) over several
workloads, frames resident, counters off; poses compared bit for bit against the first setting.
    python profiles/knob_ab.py "c2:cold:1:12 c1:cold:1:60 c2:cold:8:40" "" "SAGEICP_LOOP_PRIO=0" "SAGEICP_LOOP_PRIO=3 SAGEICP_LOOP_PRIO_LO=900"
workload spec = name:params:divisor:frames per repetition; KNOB_LIB selects a variant library; KNOB_REPS (3) repetitions."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

if os.environ.get("KNOB_LIB"):
    sage.LIB_PATH = os.path.abspath(os.environ["KNOB_LIB"])
from sage_icp_amd import synthetic as syn  # noqa: E402


def main():
    specs = sys.argv[1].split()
    settings = sys.argv[2:] or [""]
    reps = int(os.environ.get("KNOB_REPS", "3"))
    touched = set()
    for s in settings:
        for kv in s.split():
            touched.add(kv.split("=", 1)[0])
    sage.set_counting(False)
    cache = {}
    for spec in specs:
        name, params, div, K = spec.split(":")
        div, K = int(div), int(K)
        if name not in cache:
            cache.clear()
            cache[name] = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
        w = cache[name]
        p = syn.PARAMS[params]
        n = len(w["scan"]) // div
        f = sage.Frame(w["map"], w["scan"][:n])

        def run():
            return sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)

        times = {s: [] for s in settings}
        ref = None
        info = {}
        for rep in range(reps):
            for s in settings:
                for k in touched:
                    os.environ.pop(k, None)
                for kv in s.split():
                    k, v = kv.split("=", 1)
                    os.environ[k] = v
                for _ in range(2):
                    pose, st = run()
                t = time.perf_counter()
                for _ in range(K):
                    pose, st = run()
                dt = (time.perf_counter() - t) / K
                times[s].append(dt)
                if ref is None:
                    ref = pose.copy()
                info[s] = (st.iterations, st.single_launch, st.lanes_per_query, bool(np.array_equal(pose, ref)))
        print("%s %s, %d queries" % (name, params, n), flush=True)
        base = min(times[settings[0]])
        for s in settings:
            it, one, lanes, same = info[s]
            best = min(times[s])
            print("  %-64s %8.3f ms/frame (%s) %4d it %6.2f us/it  %+5.1f %%  one-launch %d lanes %d  pose %s" % (
                s or "(default)", 1e3 * best, " ".join("%.3f" % (1e3 * x) for x in times[s]), it, 1e6 * best / max(1, it),
                100.0 * (best / base - 1.0), one, lanes, "==" if same else "DIFFERS"), flush=True)


if __name__ == "__main__":
    main()
