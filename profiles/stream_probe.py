"""c3-style stream through the pipeline counterpart: per-frame time split (experiment harness)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sage_icp_amd as sage
if os.environ.get("LOOP_LIB"):           # a variant build of the library (A/B runs)
    sage.LIB_PATH = os.path.abspath(os.environ["LOOP_LIB"])
from sage_icp_amd import synthetic as syn
frames, truth = syn.make_stream(21, 40, points_per_frame=120000)
dev_update = os.environ.get("STREAM_HOST_MAP_UPDATE", "0") != "1"
p = sage.SageICP(sage.make_pipeline_config(map_update_on_device=dev_update))
print("map update on", "device" if dev_update else "host")
sage.set_profiling(2 if len(sys.argv) > 1 else 0)
kt = []
rows = []
prefetch = os.environ.get("STREAM_PREFETCH", "0") == "1"
localmap = os.environ.get("STREAM_LOCALMAP", "0") == "1"
print("LocalMap() after every frame:", "on" if localmap else "off")
print("next frame's preprocessing under this frame's ICP loop:", "on" if prefetch else "off")
frames = [np.ascontiguousarray(f, dtype=np.float64) for f in frames]
for k, f in enumerate(frames):
    t = time.perf_counter()
    if prefetch and k + 1 < len(frames):
        p.prefetch(frames[k + 1])
    pose, icp_s, tot_s, ns, st = p.RegisterFrame(f)
    wall = time.perf_counter() - t
    t_lm = 0.0
    if localmap:       # the node's default: LocalMap() after every frame (publish_frame, odometry.launch.py:23)
        t2 = time.perf_counter()
        cloud = p.LocalMap()
        t_lm = time.perf_counter() - t2
    rows.append((wall, tot_s, icp_s, ns, st.iterations, st.us_upload, t_lm))
    if st.nn_launches: kt.append((0.0, st.us_nn / st.nn_launches, 0.0, st.us_fin / st.nn_launches, st.us_wall / max(st.iterations, 1)))
r = np.array(rows[5:])
print("frames %d  source pts %.0f  iterations %.1f" % (len(r), r[:, 3].mean(), r[:, 4].mean()))
print("per frame ms: wall %.2f  (preprocess+voxelize %.2f, ICP %.2f [mirror refresh+upload %.2f], map update+rest %.2f)"
      % (1e3 * r[:, 0].mean(), 1e3 * (r[:, 1] - r[:, 2]).mean(), 1e3 * r[:, 2].mean(),
         1e-3 * r[:, 5].mean(), 1e3 * (r[:, 0] - r[:, 1]).mean()))
if localmap:
    print("LocalMap() per frame: %.2f ms for %d points (%.1f MB)  -> wall incl. LocalMap %.2f ms"
          % (1e3 * r[:, 6].mean(), len(cloud), len(cloud) * 32 / 1e6, 1e3 * (r[:, 0] + r[:, 6]).mean()))
print("map points", p.LocalMapSize() if hasattr(p, "LocalMapSize") else len(p.LocalMap()))

if kt:
    k = np.array(kt[5:]).mean(0)
    print("per iteration us (HIP events, level 2): group+probe %.1f  nn %.1f  gn %.1f  fin %.1f   | host wall per iteration %.1f" % tuple(k))
