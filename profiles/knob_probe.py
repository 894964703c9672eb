"""Time c2 frames (full and a 1/8 share) under environment knobs, one process per setting.
usage: python profiles/knob_probe.py "SAGEICP_GROUP_MAX=8" "SAGEICP_GROUP_MAX=16 SAGEICP_DEPTH=6" ..."""
import os, sys, subprocess, time
sys.path.insert(0, os.getcwd())
if "KNOB_CHILD" in os.environ:
    import sage_icp_amd as sage
    from sage_icp_amd import synthetic as syn
    if os.environ.get("KNOB_LIB"):
        sage.LIB_PATH = os.environ["KNOB_LIB"]
    w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
    p = syn.PARAMS["cold"]
    res = []
    for div in (1, 8):
        n = len(w["scan"]) // div
        f = sage.Frame(w["map"], w["scan"][:n])
        run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        for _ in range(3): run()
        t = time.perf_counter(); K = 15
        for _ in range(K): pose, st = run()
        dt = (time.perf_counter() - t) / K
        res.append("%6d q: %.3f ms %d it %.1f us/it" % (n, 1e3 * dt, st.iterations, 1e6 * dt / st.iterations))
    print("%-50s %s" % (os.environ["KNOB_CHILD"] or "(default)", " | ".join(res)), flush=True)
    sys.exit(0)
for setting in sys.argv[1:] or [""]:
    env = dict(os.environ, KNOB_CHILD=setting)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    subprocess.call([sys.executable, __file__], env=env)
