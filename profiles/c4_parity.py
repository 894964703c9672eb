"""One-off check: c4 (500k-pt scan vs 10M-pt map) pose parity of the HIP path against the CPU
oracle at full size (not part of the test suite: ~2-3 minutes of 128-core CPU time)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import oracle
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
t = time.time()
w = syn.make_workload("c4", lambda: sage.VoxelHashMap(1.0, 100.0))
om = oracle.Map(1.0, 100.0)
om.add_points(w["stream"])
print("setup %.1f s, map %d pts %d voxels, scan %d" % (time.time() - t, w["map"].size(), w["map"].num_voxels(), len(w["scan"])))
p = syn.PARAMS["steady"]
t = time.time()
pose, st = sage.register_frame(w["scan"], w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
tg = time.time() - t
t = time.time()
opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
tc = time.time() - t
e = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(opose), pose))
print("GPU %.3f s (%d iterations), CPU oracle %.1f s on %d threads (%d iterations)" % (tg, st.iterations, tc, oracle.num_threads(), ost.iterations))
print("pose delta %.3e m %.3e rad; n_corr first/last gpu %d/%d cpu %d/%d; sum candidates equal: %s"
      % (np.linalg.norm(e[:3]), np.linalg.norm(e[3:]), st.n_corr_first, st.n_corr_last, ost.n_corr_first, ost.n_corr_last,
         st.sum_candidates == ost.sum_candidates_total))
