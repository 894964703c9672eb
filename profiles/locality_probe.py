"""Would k_icp gain from a map whose regions lie in the point array in the order of their voxels
along a space-filling curve?  The same map points inserted (a) in the workload's order — regions
in first-arrival order — and (b) sorted by the Morton code of their voxel first.  Search results
are identical (the placement of regions is not observable); only where the candidates of a
neighbourhood lie in memory differs.  `python profiles/locality_probe.py c5 dense`"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
prm = syn.PARAMS[sys.argv[2] if len(sys.argv) > 2 else ("dense" if wl == "c5" else "cold")]
vs = syn.WORKLOADS[wl]["voxel"]
w = syn.make_workload(wl, lambda: sage.VoxelHashMap(vs, 100.0))


def part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def run(label, m):
    m.sync()
    f = sage.Frame(m, w["scan"])
    for _ in range(2):
        sage.register_frame(f, m, sage.IDENTITY, prm["max_dist"], prm["kernel"], prm["sem_th"])
    K = 6
    t = time.perf_counter()
    for _ in range(K):
        pose, st = sage.register_frame(f, m, sage.IDENTITY, prm["max_dist"], prm["kernel"], prm["sem_th"], return_stats=True)
    dt = (time.perf_counter() - t) / K
    print("%-28s %.3f ms/frame  %d iterations  %.1f us/iteration  (%d voxels, %d points)"
          % (label, 1e3 * dt, st.iterations, 1e6 * dt / st.iterations, m.num_voxels(), m.size()))
    return pose


stream = w["stream"]
a = run("workload order", w["map"])
vox = (stream[:, :3] / vs).astype(np.int64) + (1 << 20)
code = (part1by2(vox[:, 0]) << np.uint64(2)) | (part1by2(vox[:, 1]) << np.uint64(1)) | part1by2(vox[:, 2])
order = np.argsort(code, kind="stable")           # stable: the arrival order inside a voxel (the policy's input) is kept
m2 = sage.VoxelHashMap(vs, 100.0)
m2.AddPoints(stream[order])
b = run("regions in Morton order", m2)
print("same pose:", bool(np.array_equal(a, b)), " max |delta|", float(np.abs(a - b).max()))
