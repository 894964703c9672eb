"""Where does an iteration of the one-launch loop wait?  Timeline per workgroup from a -DSAGE_LOOP_TIMING build
(sage-icp_amd/_probe/libsageicp_looptiming.so): when it held the pose, when one of its waves took a unit beyond one
per wave and when it finished it, when the workgroup counted itself in — split by units owned and by CU population.
    python profiles/loop_tail.py [divisor 1] [cold|steady] [workload c2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import sage_icp_amd as sage  # noqa: E402

sage.LIB_PATH = os.environ.get("LOOP_LIB", os.path.join(os.path.dirname(sage.LIB_PATH), "_probe", "libsageicp_looptiming.so"))
from sage_icp_amd import synthetic as syn  # noqa: E402

div = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = syn.PARAMS[sys.argv[2] if len(sys.argv) > 2 else "cold"]
name = sys.argv[3] if len(sys.argv) > 3 else "c2"
w = syn.make_workload(name, lambda: sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0))
n = len(w["scan"]) // div
f = sage.Frame(w["map"], w["scan"][:n])
os.environ["SAGEICP_LOOP"] = "2"
sage.set_counting(False)
for _ in range(3):
    pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
assert st.single_launch == 1
IT, WG = 32, 2048
wg = np.zeros((IT, WG, 4), dtype=np.uint64)
sv = np.zeros((IT, 4), dtype=np.uint64)
sage.lib().sageicp_debug_loop_times(wg.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p))
info = np.zeros((IT, WG, 4), dtype=np.uint32)
sage.lib().sageicp_debug_loop_info(info.ctypes.data_as(C.c_void_p))
wg = wg.astype(np.float64) / 100.0
sv = sv.astype(np.float64) / 100.0
used = wg[1, :, 0] > 0
nwg = int(used.sum())
print("%s: %d queries, %d lanes/query, %d iterations, %d workgroups, %.2f us per iteration (instrumented build)"
      % (name, n, st.lanes_per_query, st.iterations, nwg, (sv[20, 3] - sv[4, 3]) / 16.0))


def q(a, x):
    return float(np.quantile(a, x)) if len(a) else float("nan")


for it in (6, 12, 20):
    if it >= min(IT, st.iterations):
        continue
    t0 = sv[it - 1, 3]                               # the pose of this iteration published
    held = wg[it - 1, used, 1] - t0
    fin = wg[it, used, 0] - t0
    s2 = wg[it, used, 2] - t0
    e2 = wg[it, used, 3] - t0
    extra = wg[it, used, 2] > 0                      # a wave of the workgroup took a second unit in this iteration
    s = sv[it] - t0
    print("iteration %d (%.2f us): solver saw all counts %.2f, sums %.2f, solved %.2f, published %.2f" % (it, s[3], s[0], s[1], s[2], s[3]))
    print("   pose held:            p0 %.2f  p50 %.2f  p100 %.2f" % (held.min(), np.median(held), held.max()))
    for nm, sel in (("workgroups WITHOUT an extra unit", ~extra), ("workgroups WITH an extra unit", extra)):
        if not sel.any():
            continue
        print("   %-34s %5d: counted in p0 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f p100 %.2f"
              % (nm, sel.sum(), fin[sel].min(), q(fin[sel], .1), q(fin[sel], .5), q(fin[sel], .9), q(fin[sel], .99), fin[sel].max()))
    if extra.any():
        print("   extra unit: taken at  p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f | lasts p10 %.2f p50 %.2f p90 %.2f p100 %.2f"
              % (s2[extra].min(), q(s2[extra], .1), q(s2[extra], .5), q(s2[extra], .9), s2[extra].max(),
                 q((e2 - s2)[extra], .1), q((e2 - s2)[extra], .5), q((e2 - s2)[extra], .9), (e2 - s2)[extra].max()))
        print("   extra unit finished -> workgroup counted in: p50 %.2f p90 %.2f p100 %.2f" % (q((fin - e2)[extra], .5), q((fin - e2)[extra], .9), (fin - e2)[extra].max()))
    # histogram of the count-in times: how many workgroups are still out at time t
    ts = np.arange(8, s[0] + 1, 2.0)
    print("   still searching at t: " + " ".join("%.0f:%d" % (t, int((fin > t).sum())) for t in ts))
    xcc = info[it, used, 0] >> 28
    cu = (info[it, used, 0] >> 8) & 0xF
    se = (info[it, used, 0] >> 13) & 0x7
    cukey = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | cu.astype(np.int64)
    keys, inv, pop = np.unique(cukey, return_inverse=True, return_counts=True)
    nx = np.zeros(len(keys))
    np.add.at(nx, inv, extra.astype(float))
    print("   %d CUs: " % len(keys) + "; ".join("%d hold %d workgroups (last counted in %.2f)" % ((pop == k).sum(), k, fin[pop[inv] == k].max()) for k in sorted(set(pop))))
    print("   extra units per CU: " + "; ".join("%d CUs with %d (last counted in %.2f)" % ((nx == k).sum(), k, fin[nx[inv] == k].max()) for k in sorted(set(nx))))
    late = np.argsort(-fin)[:10]
    print("   last ten: " + "; ".join("%.1f (%s, %d pts max, %d stale, %d pts, CU pop %d / %d extra)" % (
        fin[o], "extra %.1f-%.1f" % (s2[o], e2[o]) if extra[o] else "no extra", info[it, used, 1][o], info[it, used, 2][o], info[it, used, 3][o],
        pop[inv[o]], nx[inv[o]]) for o in late))

# ---- per wave: which SIMD, which unit it took first, when it finished it
if hasattr(sage.lib(), "sageicp_debug_loop_waves"):
    wv = np.zeros((IT, WG, 8, 2), dtype=np.uint64)
    sage.lib().sageicp_debug_loop_waves(wv.ctypes.data_as(C.c_void_p))
    for it in (12,):
        if it >= min(IT, st.iterations):
            continue
        t0 = sv[it - 1, 3]
        rec = wv[it][used]                          # [wg][wave][2]
        nwv = int((rec[0, :, 0] > 0).sum())
        end = rec[:, :nwv, 0].astype(np.float64) / 100.0 - t0
        meta = rec[:, :nwv, 1]
        start = ((rec[:, :nwv, 0] & ~np.uint64((1 << 40) - 1)) | (meta >> np.uint64(24))).astype(np.float64) / 100.0 - t0
        unit = ((meta >> np.uint64(16)) & np.uint64(0xFF)).astype(int)
        simd = ((meta >> np.uint64(4)) & np.uint64(3)).astype(int)
        slot = (meta & np.uint64(0xF)).astype(int)
        print("iteration %d, first units of the %d waves of a workgroup:" % (it, nwv))
        for k in range(nwv):
            print("   wave %d: SIMD histogram %s | slot histogram %s | unit taken histogram %s | start p50 %.2f | end p10 %.2f p50 %.2f p90 %.2f p100 %.2f"
                  % (k, np.bincount(simd[:, k], minlength=4).tolist(), np.bincount(slot[:, k], minlength=10).tolist(),
                     np.bincount(unit[:, k], minlength=nwv).tolist(), np.median(start[:, k]), q(end[:, k], .1), q(end[:, k], .5), q(end[:, k], .9), end[:, k].max()))
        for u in range(nwv):
            sel = unit == u
            print("   unit %d (by last iteration's work, heaviest first): lasts p10 %.2f p50 %.2f p90 %.2f p100 %.2f | ends p50 %.2f p90 %.2f p100 %.2f"
                  % (u, q((end - start)[sel], .1), q((end - start)[sel], .5), q((end - start)[sel], .9), (end - start)[sel].max(),
                     q(end[sel], .5), q(end[sel], .9), end[sel].max()))
        print("   distinct SIMDs per workgroup: " + str(np.bincount([len(set(r)) for r in simd.tolist()], minlength=5).tolist()))
        # per SIMD of the machine: which units its waves took, when its last first-unit ended
        xcc = (info[it, used, 0] >> 28).astype(np.int64)
        cu = ((info[it, used, 0] >> 8) & 0xF).astype(np.int64)
        se = ((info[it, used, 0] >> 13) & 0x7).astype(np.int64)
        key = (((xcc << 4 | se) << 4 | cu)[:, None] << 2) | simd
        keys, inv = np.unique(key.ravel(), return_inverse=True)
        heavy = np.zeros(len(keys)); cnt = np.zeros(len(keys)); last = np.zeros(len(keys))
        np.add.at(heavy, inv, (unit.ravel() == 0).astype(float))
        np.add.at(cnt, inv, 1.0)
        np.maximum.at(last, inv, end.ravel())
        print("   %d SIMDs: waves per SIMD histogram %s; heaviest units (unit 0) per SIMD histogram %s" % (len(keys), np.bincount(cnt.astype(int)).tolist(), np.bincount(heavy.astype(int)).tolist()))
        for h in sorted(set(heavy.astype(int))):
            print("      SIMDs with %d heaviest units: %d, last first-unit ends mean %.2f max %.2f" % (h, (heavy == h).sum(), last[heavy == h].mean(), last[heavy == h].max()))
