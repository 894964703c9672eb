cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
(timeout 600 python profiles/phase_probe.py 1 8 > gpurun_out/r02c/phase.txt 2>&1); grep -v "^k_gn\|^   pair\|^   block\|^   rel\|^   fin" gpurun_out/r02c/phase.txt
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
for pn in ("cold", "steady"):
    p = syn.PARAMS[pn]
    f = sage.Frame(w["map"], w["scan"])
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    print(pn, st.iterations, "pairs/cand", st.pairs_evaluated / st.sum_candidates)
    # prune ratio at the converged pose: one more registration from the result
    pose2, st2 = sage.register_frame(f, w["map"], pose, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    print(pn, "from converged pose:", st2.iterations, "pairs/cand", st2.pairs_evaluated / st2.sum_candidates)
PY
