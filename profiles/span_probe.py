"""Timeline of one k_icp launch (instrumented build, -DSAGE_NN_TIMING prebuilt as SPAN_LIB): when
every wave starts and ends, on which XCD / CU, and how the number of resident waves decays.
usage: SPAN_LIB=variants/tNN.so python profiles/span_probe.py [iteration ...] [--div N]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
sage.LIB_PATH = os.environ["SPAN_LIB"]
from sage_icp_amd import synthetic as syn
L = sage.lib()
args = sys.argv[1:]
div = 1
if "--div" in args:
    k = args.index("--div"); div = int(args[k + 1]); del args[k:k + 2]
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
n = len(w["scan"]) // div
f = sage.Frame(w["map"], w["scan"][:n])
run = lambda: sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
run()
for it in [int(a) for a in args] or [0, 20, 100]:
    L.sageicp_debug_nn_spans(None, 0, it)
    pose, st = run()
    nw = (n + (64 // st.lanes_per_query) - 1) // (64 // st.lanes_per_query)
    buf = (C.c_ulonglong * (4 * nw))()
    L.sageicp_debug_nn_spans(buf, nw, it)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 4).astype(np.int64)
    t0 = a[:, 0].min()
    s, e = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # us
    hw = a[:, 2]
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
    print("iteration %d: %d waves, %d lanes/query; first start 0, last start %.1f us, last end %.1f us" % (it, nw, st.lanes_per_query, s.max(), e.max()))
    life = e - s
    print("   lifetime us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
    ts = np.arange(0, e.max() + 2, 2.0)
    print("   resident waves at t (us): " + " ".join("%d:%d" % (t, ((s <= t) & (e > t)).sum()) for t in ts))
    late = np.argsort(-e)[:12]
    print("   last waves to end: " + " ".join("w%d[%.0f-%.0f p%d]" % (i, s[i], e[i], a[i, 3] >> 32) for i in late))
    xp = a[:, 2]
    steps, exact, lanes = xp >> 40, (xp >> 20) & 0xFFFFF, xp & 0xFFFFF
    print("   pair steps per wave %.1f, of them with a fetch of full records %.2f (%.0f %%), lanes fetching per such step %.1f"
          % (steps.mean(), exact.mean(), 100.0 * exact.sum() / max(steps.sum(), 1), lanes.sum() / max(exact.sum(), 1)))
    xcc = (a[:, 2] >> 32) & 15
    key = xcc * 1024 + se * 32 + sh * 16 + cu
    ends = {}
    for kk, ee in zip(key, e): ends[kk] = max(ends.get(kk, 0), ee)
    ev = np.array(sorted(ends.values()))
    if False: print("   %d CUs seen; a CU's last wave ends at: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (len(ev), ev[0], *np.percentile(ev, [10, 50, 90]), ev[-1]))
    if False: print("   per XCD: waves " + " ".join("%d" % (xcc == x).sum() for x in range(8)) + " | last end " + " ".join("%.0f" % e[xcc == x].max() for x in range(8) if (xcc == x).any()) + " | mean life " + " ".join("%.1f" % life[xcc == x].mean() for x in range(8) if (xcc == x).any()))
    mx, sm = a[:, 3] >> 32, a[:, 3] & 0xFFFFFFFF
    qw = 64 // st.lanes_per_query
    print("   points handed out per query: mean %.1f; per wave max-over-queries: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %d  (lockstep efficiency %.2f)"
          % (sm.sum() / n, mx.mean(), *np.percentile(mx, [50, 90, 99]), mx.max(), sm.sum() / max((mx * qw).sum(), 1)))
    c = np.corrcoef(mx, life)[0, 1]
    print("   correlation(wave max points, wave lifetime) = %.2f; lifetime by max-points quintile: " % c + " ".join("%.1f" % life[np.argsort(mx)[i * nw // 5:(i + 1) * nw // 5]].mean() for i in range(5)))

# which phase makes a slow wave slow: per-slot means over a whole registration, by lifetime decile
L.sageicp_debug_nn_phases((C.c_ulonglong * 16)(), 1)
pose, st = run()
nw = (n + (64 // st.lanes_per_query) - 1) // (64 // st.lanes_per_query)
raw = (C.c_ulonglong * (8 * nw))()
L.sageicp_debug_nn_raw(raw, nw)
r = np.frombuffer(raw, dtype=np.uint64).reshape(nw, 8).astype(np.float64)
r = r[r[:, 7] > 0]
m = r[:, :6] / r[:, 7:8]
order = np.argsort(m[:, 5])
print("per-wave-slot means over %d iterations, by lifetime decile (shader cycles): loads | row | seed+home | neighbours | epilogue | lifetime" % st.iterations)
for d in range(10):
    sel = order[d * len(order) // 10:(d + 1) * len(order) // 10]
    print("   decile %d: " % d + " ".join("%7.0f" % v for v in m[sel].mean(axis=0)))
