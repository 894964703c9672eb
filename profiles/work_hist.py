"""Distribution of the per-query scan work of k_icp (map points handed to a query) at the end of a c2
registration, and what wave-level lockstep costs under a few groupings (instrumented build).
usage: SPAN_LIB=variants/tNN.so python profiles/work_hist.py"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import sage_icp_amd as sage
sage.LIB_PATH = os.environ["SPAN_LIB"]
from sage_icp_amd import synthetic as syn
L = sage.lib()
w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0))
p = syn.PARAMS["cold"]
n = len(w["scan"])
f = sage.Frame(w["map"], w["scan"])
pose, st = sage.register_frame(f, w["map"], sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"], return_stats=True)
buf = (C.c_uint32 * n)()
L.sageicp_debug_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc = L.sageicp_debug_work(w["map"]._h, buf, n)
r = np.frombuffer(buf, dtype=np.uint32).astype(np.int64)
print("rc", rc, "queries", n, "points handed out per query: mean %.1f" % r.mean(), "quantiles 50/75/90/95/99/99.9/max:", *np.percentile(r, [50, 75, 90, 95, 99, 99.9]).round(0), r.max())
for T in (48, 64, 96, 128, 192, 256):
    print("   > %3d points: %5.2f %% of the queries, %5.1f %% of the points" % (T, 100.0 * (r > T).mean(), 100.0 * r[r > T].sum() / r.sum()))
QW = 16
m = (n // QW) * QW
g = r[:m].reshape(-1, QW)
print("waves of %d consecutive queries: sum of wave maxima / sum of points = %.2f (lockstep efficiency %.2f)" % (QW, g.max(1).sum() * QW / r[:m].sum(), r[:m].sum() / (g.max(1).sum() * QW)))
for chunk in (64, 256, 1024, 4096):
    mm = (n // chunk) * chunk
    s = np.sort(r[:mm].reshape(-1, chunk), axis=1).reshape(-1, QW)
    print("   sorted by work inside chunks of %4d consecutive queries: efficiency %.2f, wave max mean %.1f p99 %.0f" % (chunk, r[:mm].sum() / (s.max(1).sum() * QW), s.max(1).mean(), np.percentile(s.max(1), 99)))
for T in (64, 96, 128):
    capped = np.minimum(g, T).max(1)
    heavy = (g > T)
    coop = (np.ceil(np.where(heavy, g, 0) / 64.0)).sum(1)
    print("   cap %3d + wave-cooperative rest: per-lane steps (points/lane) mean %.1f (now %.1f); cooperative rounds per wave mean %.2f, heavy queries per wave %.2f"
          % (T, capped.mean() / 4, g.max(1).mean() / 4, coop.mean(), heavy.sum(1).mean()))
