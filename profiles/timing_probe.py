"""Per-phase s_memtime totals of k_nn waves (instrumented build, -DSAGE_NN_TIMING)."""
import os, sys, subprocess, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
# build an instrumented copy of the library
src=["sage-icp_amd/csrc/%s.hip" % n for n in ("kernels", "sort", "preprocess", "map_update", "capi")]
out="gpurun_out/libsageicp_timing.so"
os.makedirs("gpurun_out", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-shared","-ffp-contract=off","-DSAGE_NN_TIMING"]+src+["-o",out,"-ldl"])
import sage_icp_amd as sage
sage.LIB_PATH=os.path.abspath(out)
sage._lib=None
from sage_icp_amd import synthetic as syn
L=sage.lib()
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
f = sage.Frame(w["map"], w["scan"])
p = syn.PARAMS["cold"]
sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
buf=(C.c_ulonglong*8)()
L.sageicp_debug_nn_phases(buf, 1)
pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
L.sageicp_debug_nn_phases(buf, 0)
v=list(buf); waves=v[4]; launches=st.iterations
print("iterations", launches, "waves", waves, "waves/launch", waves/launches)
names=["prologue / group header","enumerate","nn_group(pairs+reduce+store)","wave lifetime"]
for i in range(4): print("%-32s %10.0f ticks/wave/launch"%(names[i], v[i]/waves))
print("max wave lifetime ticks", v[5], "(s_memtime ticks: 100 MHz const clock => 10 ns each?)")
