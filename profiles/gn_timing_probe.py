"""Critical-path phases of k_gn's last workgroup (s_memrealtime, 10 ns ticks); instrumented build."""
import os, sys, subprocess, ctypes as C
sys.path.insert(0, os.getcwd())
src = ["sage-icp_amd/csrc/%s.hip" % n for n in ("kernels", "sort", "preprocess", "map_update", "capi")]
out = "gpurun_out/libsageicp_gntiming.so"
os.makedirs("gpurun_out", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-DSAGE_GN_TIMING"] + src + ["-o", out, "-ldl"])
import sage_icp_amd as sage
sage.LIB_PATH = os.path.abspath(out)
sage._lib = None
from sage_icp_amd import synthetic as syn
L = sage.lib()
for name in sys.argv[1:] or ["c1", "c2"]:
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
    f = sage.Frame(w["map"], w["scan"]); p = syn.PARAMS["cold"]
    sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    buf = (C.c_ulonglong * 16)()
    L.sageicp_debug_gn_phases(buf, 1)
    sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    L.sageicp_debug_gn_phases(buf, 0)
    v = list(buf); n = max(v[4], 1)
    names = ["pair loop", "wave+block reduce, partial store", "release+ticket+acquire", "finish_iteration (all)"]
    print(name, "launches", n)
    for i in range(4): print("  %-36s %6.2f us" % (names[i], v[i] / n / 100.0))
    fn = ["  fin: reduce partials", "  fin: assemble + LDLT", "  fin: se3 exp", "  fin: compose + norm + state"]
    for i in range(4): print("  %-36s %6.2f us" % (fn[i], v[8 + i] / n / 100.0))
