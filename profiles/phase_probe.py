"""Per-phase device clocks of k_nn waves and of k_gn's last workgroup for a share of the c2 frame
(instrumented builds, -DSAGE_NN_TIMING / -DSAGE_GN_TIMING; one process per build).
usage: python profiles/phase_probe.py [divisors ...]      e.g. 1 8"""
import os, sys, subprocess, ctypes as C
sys.path.insert(0, os.getcwd())
SRC = ["sage-icp_amd/csrc/%s.hip" % n for n in ("kernels", "sort", "preprocess", "map_update", "capi")]
FLAGS = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
if "PHASE_LIB" not in os.environ:
    os.makedirs("gpurun_out", exist_ok=True)
    for kind in ("NN", "GN"):
        out = os.path.abspath("gpurun_out/libsageicp_t%s.so" % kind)
        subprocess.check_call(FLAGS + ["-DSAGE_%s_TIMING" % kind] + SRC + ["-o", out, "-ldl"])
        subprocess.call([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, PHASE_LIB=out, PHASE_KIND=kind))
    sys.exit(0)
import sage_icp_amd as sage
sage.LIB_PATH = os.environ["PHASE_LIB"]
from sage_icp_amd import synthetic as syn
L = sage.lib()
kind = os.environ["PHASE_KIND"]
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
for div in [int(a) for a in sys.argv[1:]] or [1, 8, 120]:
    n = len(w["scan"]) // div
    f = sage.Frame(w["map"], w["scan"][:n])
    run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    run()
    if kind == "NN":
        buf = (C.c_ulonglong * 16)()
        L.sageicp_debug_nn_phases(buf, 1)
        pose, st = run()
        L.sageicp_debug_nn_phases(buf, 0)
        v = list(buf); waves = max(v[7], 1)
        print("k_icp %d queries, %d lanes/query: %d iterations, %.0f waves/launch, pairs evaluated / candidates = %.3f"
              % (n, st.lanes_per_query, st.iterations, waves / st.iterations, st.pairs_evaluated / max(st.sum_candidates, 1)))
        for i, nm in enumerate(["frame + row key loads, home voxel", "row staging / rebuild, gaps", "seed + home voxel scan",
                                "bound, need mask, neighbour scan, argmin", "epilogue (GN terms, reductions)"]):
            print("   %-44s %8.0f shader cycles per wave" % (nm, v[i] / waves))
        print("   %-44s %8.0f shader cycles = %.2f us  (clock %.2f GHz)" % ("wave lifetime", v[5] / waves, v[6] / waves / 100.0, v[5] / max(v[6], 1) / 10.0))
        print("   slowest wave slot: mean lifetime %d cycles" % v[8])
    else:
        buf = (C.c_ulonglong * 16)()
        L.sageicp_debug_gn_phases(buf, 1)
        run()
        L.sageicp_debug_gn_phases(buf, 0)
        v = list(buf); k = max(v[4], 1)
        print("k_fin after %d queries, %d launches (100-MHz ticks -> us)" % (n, k))
        for i, nm in enumerate(["reduce the partials", "assemble + LDLT", "se3 exp", "compose + norm + state + progress"]):
            print("   %-36s %6.2f us" % (nm, v[8 + i] / k / 100.0))
