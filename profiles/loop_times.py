"""Timeline of the one-launch loop (k_loop) from a build with -DSAGE_LOOP_TIMING: per iteration, when the
workgroups counted themselves in, when the solving wave saw all counts / had the sums / had the step /
had published, and when the workgroups held the next pose.  100-MHz ticks -> microseconds.
    python profiles/loop_times.py [divisor of the frame, default 8] [cold|steady] [workload, default c2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

sage.LIB_PATH = os.environ.get("LOOP_LIB", os.path.join(os.path.dirname(sage.LIB_PATH), "_probe", "libsageicp_looptiming.so"))
from sage_icp_amd import synthetic as syn  # noqa: E402

div = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = syn.PARAMS[sys.argv[2] if len(sys.argv) > 2 else "cold"]
name = sys.argv[3] if len(sys.argv) > 3 else "c2"
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
n = len(w["scan"]) // div
f = sage.Frame(w["map"], w["scan"][:n])
os.environ["SAGEICP_LOOP"] = "2"
for _ in range(3):
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
assert st.single_launch == 1
IT, WG = 32, 2048
wg = np.zeros((IT, WG, 4), dtype=np.uint64)      # (round 6: four stamps per workgroup; this script reads the first two — loop_tail.py the rest)
sv = np.zeros((IT, 4), dtype=np.uint64)
sage.lib().sageicp_debug_loop_times(wg.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p))
wg = wg.astype(np.float64) / 100.0
sv = sv.astype(np.float64) / 100.0
used = wg[1, :, 0] > 0
print("%d queries, %d lanes/query, %d iterations, %d query workgroups" % (n, st.lanes_per_query, st.iterations, used.sum()))
print("per iteration, us after the previous pose was published:")
print("  it | counted in: first  median  p90   last | solver: all in   sums   solved  published | pose held: first median last | iteration")
rows = []
for it in range(1, min(IT, st.iterations)):
    t0 = sv[it - 1, 3]
    a = np.sort(wg[it, used, 0]) - t0
    h = np.sort(wg[it, used, 1]) - t0
    s = sv[it] - t0
    prev_h = np.sort(wg[it - 1, used, 1]) - t0
    rows.append((prev_h[len(prev_h) // 2], a[0], a[len(a) // 2], a[int(len(a) * 0.9)], a[-1], s[0], s[1], s[2], s[3], h[0], h[len(h) // 2], h[-1]))
    if it < 6 or it % 16 == 0:
        print("  %3d |          %6.2f %6.2f %6.2f %6.2f |        %6.2f %6.2f %6.2f %6.2f |         %6.2f %6.2f %6.2f | %6.2f"
              % (it, a[0], a[len(a) // 2], a[int(len(a) * 0.9)], a[-1], s[0], s[1], s[2], s[3], h[0] - s[3], h[len(h) // 2] - s[3], h[-1] - s[3], s[3]))
r = np.array(rows[2:]).mean(0)
print("mean: pose held (median workgroup, previous iteration) %.2f | counted in first %.2f median %.2f p90 %.2f last %.2f | all in %.2f sums %.2f solved %.2f published %.2f"
      % tuple(r[:9]))
print("      -> wait for the pose %.2f, search of the median workgroup %.2f, of the last %.2f, last count -> seen %.2f, read sums %.2f, solve %.2f, publish %.2f"
      % (r[0], r[2] - r[0], r[4] - r[0], r[5] - r[4], r[6] - r[5], r[7] - r[6], r[8] - r[7]))

# ---- which workgroups are the slow ones?
info = np.zeros((IT, WG, 4), dtype=np.uint32)
if hasattr(sage.lib(), "sageicp_debug_loop_info"):
    sage.lib().sageicp_debug_loop_info(info.ctypes.data_as(C.c_void_p))
    its = [k for k in (8, 24, 40) if k < min(IT, st.iterations)]
    for it in its:
        t0 = sv[it - 1, 3]
        arr = wg[it, used, 0] - t0
        held = wg[it - 1, used, 1] - t0
        mx, stl, sm = info[it, used, 1].astype(float), info[it, used, 2].astype(float), info[it, used, 3].astype(float)
        xcc = info[it, used, 0] >> 28
        cu = (info[it, used, 0] >> 8) & 0xF
        se = (info[it, used, 0] >> 13) & 0x7
        dur = arr - held
        print("iteration %d: search time of a workgroup (pose held -> counted in): mean %.2f  p50 %.2f  p90 %.2f  max %.2f"
              % (it, dur.mean(), np.median(dur), np.quantile(dur, 0.9), dur.max()))
        print("   correlation with: max points of one query %.2f | points of the workgroup %.2f | stale queries %.2f"
              % (np.corrcoef(dur, mx)[0, 1], np.corrcoef(dur, sm)[0, 1], np.corrcoef(dur, stl)[0, 1] if stl.std() > 0 else 0.0))
        print("   per XCD: mean search time " + " ".join("%.2f" % dur[xcc == x].mean() for x in range(8) if (xcc == x).any())
              + " | points " + " ".join("%.0f" % sm[xcc == x].sum() for x in range(8) if (xcc == x).any()))
        order = np.argsort(-dur)[:8]
        print("   slowest: " + "; ".join("%.1f us (xcd %d se %d cu %d, max %d pts, %d stale, %d pts)" % (dur[o], xcc[o], se[o], cu[o], mx[o], stl[o], sm[o]) for o in order))
        cukey = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | cu.astype(np.int64)
        keys, inv, pop = np.unique(cukey, return_inverse=True, return_counts=True)
        print("   %d CUs hold the %d workgroups: " % (len(keys), used.sum()) + "; ".join(
            "%d CUs with %d workgroups: search time mean %.2f, last %.2f" % ((pop == k).sum(), k, dur[pop[inv] == k].mean(), dur[pop[inv] == k].max())
            for k in sorted(set(pop))))
        for lo_, hi_ in ((0, 0), (1, 1), (2, 3), (4, 7), (8, 15), (16, 10 ** 6)):
            sel = (stl >= lo_) & (stl <= hi_)
            if sel.any():
                print("   %5d workgroups with %d..%d stale queries: search time mean %.2f p90 %.2f max %.2f"
                      % (sel.sum(), lo_, min(hi_, 999), dur[sel].mean(), np.quantile(dur[sel], 0.9), dur[sel].max()))
        q = np.quantile(sm, [0.2, 0.4, 0.6, 0.8])
        b = np.digitize(sm, q)
        print("   search time by quintile of the workgroup's points: " + " ".join("%.2f" % dur[b == k].mean() for k in range(5)))

# ---- where a wave's iteration goes (cycles per phase, mean over all waves and iterations)
if hasattr(sage.lib(), "sageicp_debug_loop_phases"):
    ph = np.zeros(16, dtype=np.uint64)
    sage.lib().sageicp_debug_loop_phases(ph.ctypes.data_as(C.c_void_p), 1)
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
    sage.lib().sageicp_debug_loop_phases(ph.ctypes.data_as(C.c_void_p), 1)
    k = float(ph[10]) or 1.0
    names = ["pose, query, home voxel", "row rebuild (stale), face gaps", "seed, home-only scan (unseeded)", "bounds -> need mask",
             "scan", "argmin", "answer's record (if changed)", "pair terms, wave reduction, ticket"]
    ghz = 2.1
    print("mean over %d wave-iterations, us at %.1f GHz (s_memtime cycles):" % (int(k), ghz))
    for i, nm in enumerate(names):
        print("   %-38s %6.2f" % (nm, ph[i] / k / ghz / 1e3))
    print("   %-38s %6.2f   (sum of the above)" % ("body", ph[:8].sum() / k / ghz / 1e3))
    print("   %-38s %6.2f   (last wave of a workgroup only: sums -> accumulators, atomics acknowledged, count)" % ("closing the workgroup", ph[9] / k / ghz / 1e3 * (st.lanes_per_query and 1)))
    print("   %-38s %6.2f   (barrier: the workgroup's poller has the next pose)" % ("waiting", ph[8] / k / ghz / 1e3))
