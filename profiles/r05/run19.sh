mkdir -p gpurun_out
timeout 600 python profiles/loop_times.py 1 cold c1 > gpurun_out/r05_run19_times_c1.txt 2>&1
head -16 gpurun_out/r05_run19_times_c1.txt; tail -14 gpurun_out/r05_run19_times_c1.txt
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c1", lambda: sage.VoxelHashMap(0.8, 100.0))
p = syn.PARAMS["cold"]
f = sage.Frame(w["map"], w["scan"])
for env in ({"SAGEICP_LOOP": "0"}, {"SAGEICP_LOOP": "1"}, {"SAGEICP_LOOP": "1", "SAGEICP_MAX_ITER": "1"}, {"SAGEICP_LOOP": "0", "SAGEICP_MAX_ITER": "1"}):
    for k in ("SAGEICP_LOOP", "SAGEICP_MAX_ITER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(5):
        sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    t = time.perf_counter()
    for _ in range(50):
        pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    dt = (time.perf_counter() - t) / 50
    print(env, "%.1f us/frame, %d iterations, one_launch=%d, %d lanes" % (1e6 * dt, st.iterations, st.single_launch, st.lanes_per_query))
PY
