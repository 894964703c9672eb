"""k_icp (launch per iteration) from a -DSAGE_NN_TIMING build: per-phase cycles of a wave and the occupancy timeline of
one launch (when its waves started and ended) — where a launch of a frame beyond the LDS (c4) spends its time.
    python profiles/icp_tail.py [workload c4] [params steady] [iteration 20]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

sage.LIB_PATH = os.environ.get("NN_LIB", os.path.join(os.path.dirname(sage.LIB_PATH), "_probe", "libsageicp_nntiming.so"))
from sage_icp_amd import synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
p = syn.PARAMS[sys.argv[2] if len(sys.argv) > 2 else "steady"]
it_span = int(sys.argv[3]) if len(sys.argv) > 3 else 20
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
f = sage.Frame(w["map"], w["scan"])
os.environ["SAGEICP_LOOP"] = "0"
sage.set_counting(False)
L = sage.lib()
run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)  # noqa: E731
run()
buf = (C.c_ulonglong * 16)()
L.sageicp_debug_nn_phases(buf, 1)
L.sageicp_debug_nn_spans(None, C.c_uint(0), C.c_int(it_span))
pose, st = run()
L.sageicp_debug_nn_phases(buf, 0)
v = list(buf)
waves = max(v[7], 1)
nw = int(round(waves / st.iterations))
print("%s: %d queries, %d lanes/query, %d iterations, %d waves per launch" % (name, len(w["scan"]), st.lanes_per_query, st.iterations, nw))
for i, nm in enumerate(["frame + row key loads, home voxel", "row staging / rebuild, gaps", "seed + home voxel scan",
                        "bound, need mask, neighbour scan, argmin", "epilogue (GN terms, reductions)"]):
    print("   %-44s %8.0f shader cycles per wave" % (nm, v[i] / waves))
print("   %-44s %8.0f shader cycles = %.2f us  (clock %.2f GHz)" % ("wave lifetime", v[5] / waves, v[6] / waves / 100.0, v[5] / max(v[6], 1) / 10.0))
sp = np.zeros((min(nw, 1 << 17), 4), dtype=np.uint64)
L.sageicp_debug_nn_spans(sp.ctypes.data_as(C.c_void_p), C.c_uint(len(sp)), C.c_int(it_span))
t0 = float(sp[:, 0].min())
s = (sp[:, 0].astype(np.float64) - t0) / 100.0
e = (sp[:, 1].astype(np.float64) - t0) / 100.0
print("launch of iteration %d: first wave starts 0.00, last wave ends %.2f us; a wave lasts p10 %.2f p50 %.2f p90 %.2f max %.2f us"
      % (it_span, e.max(), np.quantile(e - s, .1), np.median(e - s), np.quantile(e - s, .9), (e - s).max()))
ts = np.arange(0, e.max() + 4, 4.0)
print("   waves in flight at t: " + " ".join("%.0f:%d" % (t, int(((s <= t) & (e > t)).sum())) for t in ts))
print("   waves started by t:   " + " ".join("%.0f:%d" % (t, int((s <= t).sum())) for t in ts))
mx = (sp[:, 3] >> np.uint64(32)).astype(np.float64)
print("   max points handed to a query of the wave: p10 %.0f p50 %.0f p90 %.0f max %.0f; correlation with the wave's duration %.2f"
      % (np.quantile(mx, .1), np.median(mx), np.quantile(mx, .9), mx.max(), np.corrcoef(mx, e - s)[0, 1]))
late = np.argsort(-e)[:8]
print("   last to end: " + "; ".join("%.1f (started %.1f, %d pts max)" % (e[o], s[o], mx[o]) for o in late))
