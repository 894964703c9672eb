# epilogue from the operands (transposition first) against the terms-first epilogue: same bits? faster?
mkdir -p gpurun_out/r06
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/r06/pairs_first_bits.txt
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
CH = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
out = {}
for name, params, div in [("c2","cold",1),("c2","steady",4),("c1","cold",1),("c5","dense",1),("c2","cold",16)]:
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
    p = syn.PARAMS[params]; n = len(w["scan"]) // div
    f = sage.Frame(w["map"], w["scan"][:n])
    for loop in ("1", "0"):
        os.environ["SAGEICP_LOOP"] = loop
        pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
        print(name, params, n, "loop", loop, st.iterations, pose.tobytes().hex())
'''
a = subprocess.run([sys.executable, "-c", CH], capture_output=True, text=True, env=dict(os.environ, SAGEICP_VARIANT_LIB="sage-icp_amd/_probe/libsageicp_prev.so")).stdout
b = subprocess.run([sys.executable, "-c", CH], capture_output=True, text=True).stdout
for la, lb in zip(a.splitlines(), b.splitlines()):
    print(la[:40], "SAME" if la == lb else "DIFFERENT")
print(len(a.splitlines()), len(b.splitlines()))
PY
AB_WORKLOADS="c2:cold:12 c1:cold:60 c4:steady:4 c5:dense:10" timeout 1500 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_prev.so product 2>&1 | tee gpurun_out/r06/pairs_first_ab.txt
