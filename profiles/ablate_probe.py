"""Ablation timing of k_nn on the c2 workload (experiment harness, not part of the product)."""
import os, sys, subprocess, time, numpy as np
sys.path.insert(0, os.getcwd())
variants = {"full": [], "noeval": ["-DSAGE_ABLATE_NOEVAL"], "noload": ["-DSAGE_ABLATE_NOLOAD"],
            "noload_noeval": ["-DSAGE_ABLATE_NOLOAD", "-DSAGE_ABLATE_NOEVAL"]}
extra = sys.argv[1:]
src=["sage-icp_amd/csrc/kernels.hip","sage-icp_amd/csrc/sort.hip","sage-icp_amd/csrc/capi.hip"]
os.makedirs("gpurun_out", exist_ok=True)
import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn
import importlib
first = True
for name, flags in variants.items():
    out=os.path.abspath("gpurun_out/lib_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-shared","-ffp-contract=off"]+flags+extra+src+["-o",out,"-ldl"])
    # fresh process per variant keeps library state clean
    code = f"""
import sys, os; sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
sage.LIB_PATH={out!r}; sage._lib=None
from sage_icp_amd import synthetic as syn
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
f = sage.Frame(w["map"], w["scan"]); p = syn.PARAMS["cold"]
sage.set_profiling(True)
tot=0; n=0; it=0
for r in range(3):
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    if r: tot+=st.us_nn; n+=st.nn_launches; it=st.iterations
print({name!r}, "k_nn avg us", round(tot/max(n,1),1), "launches", n, "iters", it, "us_group", round(st.us_group/max(st.nn_launches,1),1))
"""
    subprocess.call([sys.executable, "-c", code])
