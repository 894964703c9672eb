"""Phase stamps of k_fin (s_memrealtime, 100-MHz ticks) from an instrumented build
(-DSAGE_GN_TIMING, built in-tree beforehand: sage-icp_amd/_probe/libsageicp_tGN.so, see
profiles/r03/run*.sh) on c2 / c4 at full size.
usage: python profiles/fin_phases.py [c2 c4 ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage

sage.LIB_PATH = os.path.abspath(os.environ.get("PHASE_LIB", "sage-icp_amd/_probe/libsageicp_tGN.so"))
from sage_icp_amd import synthetic as syn

L = sage.lib()
for wl in sys.argv[1:] or ["c2", "c4"]:
    cfg = syn.WORKLOADS[wl]
    w = syn.make_workload(wl, lambda: sage.VoxelHashMap(cfg["voxel"], 100.0))
    p = syn.PARAMS["steady" if wl == "c4" else "cold"]
    f = sage.Frame(w["map"], w["scan"])
    run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],
                                      return_stats=True)
    run()
    buf = (C.c_ulonglong * 16)()
    L.sageicp_debug_gn_phases(buf, 1)
    pose, st = run()
    L.sageicp_debug_gn_phases(buf, 0)
    v = list(buf)
    k = max(v[4], 1)
    print("k_fin, %s (%d queries, %d lanes/query, %d iterations): %d launches stamped (100-MHz ticks -> us)"
          % (wl, len(w["scan"]), st.lanes_per_query, st.iterations, k))
    names = ["reduce the partials (launch -> sums in LDS)", "assemble + LDLT", "se3 exp",
             "compose + norm + state + progress"]
    tot = 0.0
    for i, nm in enumerate(names):
        us = v[8 + i] / k / 100.0
        tot += us
        print("   %-46s %6.2f us" % (nm, us))
    print("   %-46s %6.2f us" % ("sum (first instruction -> progress word)", tot))
    del f, w
