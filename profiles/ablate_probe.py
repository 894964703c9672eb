"""Build-variant timing of k_nn on the c2 workload (experiment harness, not part of the product).
usage: python profiles/ablate_probe.py name1:-DFLAG1,-DFLAG2 name2: ..."""
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
variants = {}
for a in sys.argv[1:]:
    name, _, flags = a.partition(":")
    variants[name] = [f for f in flags.split(",") if f]
src=["sage-icp_amd/csrc/kernels.hip","sage-icp_amd/csrc/sort.hip","sage-icp_amd/csrc/capi.hip"]
os.makedirs("gpurun_out", exist_ok=True)
for name, flags in variants.items():
    out=os.path.abspath("gpurun_out/lib_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-shared","-ffp-contract=off"]+flags+src+["-o",out,"-ldl"])
    code = f"""
import sys, os; sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
sage.LIB_PATH={out!r}; sage._lib=None
from sage_icp_amd import synthetic as syn
import time
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
f = sage.Frame(w["map"], w["scan"]); p = syn.PARAMS["cold"]
sage.set_profiling(1)
tot=0; n=0; it=0; wall=0
for r in range(4):
    t=time.perf_counter()
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    dt=time.perf_counter()-t
    if r: tot+=st.us_nn; n+=st.nn_launches; it=st.iterations; wall+=dt
print({name!r}, "k_nn avg us", round(tot/max(n,1),1), "iters", it, "ms/frame", round(wall/3*1e3,2))
"""
    subprocess.call([sys.executable, "-c", code])
