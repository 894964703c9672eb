"""Device memory one workload's voxel map takes, as the runtime reports it (hipMemGetInfo through
torch): `python profiles/map_memory.py [workload ...]`.  SAGEICP_SIZE_CLASSES=0 gives every voxel a
block of max_points_per_voxel points (the layout of rounds 1-2) for comparison."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import sage_icp_amd as sage
from sage_icp_amd import synthetic as syn


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


def lattice():
    """the 10.9 M-voxel 0.1 m test map of tests/test_gpu_parity.py: one point per voxel"""
    import numpy as np
    n = 222
    g = (np.arange(n, dtype=np.float64) - n // 2) * 0.1 + 0.031
    mp = np.empty((n * n * n, 4))
    mp[:, 0] = np.repeat(g, n * n)
    mp[:, 1] = np.tile(np.repeat(g, n), n)
    mp[:, 2] = np.tile(g, n * n)
    mp[:, 3] = 40
    torch.zeros(1, device="cuda")
    base = used()
    m = sage.VoxelHashMap(0.1, 100.0)
    m.AddPoints(mp)
    m.sync()
    print("lattice: %d voxels, %d points, %d point slots (%.2f GB); device memory: map mirror %.2f GB (size classes %s)"
          % (m.num_voxels(), m.size(), m.point_slots(), m.point_slots() * 32 / 1e9, (used() - base) / 1e9,
             os.environ.get("SAGEICP_SIZE_CLASSES", "on")))


for name in (sys.argv[1:] or ["c5"]):
    if name == "lattice":
        lattice()
        continue
    wl = syn.WORKLOADS[name]
    torch.zeros(1, device="cuda")
    base = used()
    w = syn.make_workload(name, lambda: sage.VoxelHashMap(wl["voxel"], 100.0, device=0))
    vmap = w["map"]
    vmap.sync()
    m_sync = used()
    scan = w["scan"]
    prm = syn.PARAMS["dense" if name == "c5" else "cold"]
    sage.register_frame(scan, vmap, sage.IDENTITY, prm["max_dist"], prm["kernel"], prm["sem_th"])
    m_reg = used()
    print("%s: %d voxels, %d points; device memory: map mirror %.2f GB, after one RegisterFrame %.2f GB "
          "(size classes %s)" % (name, vmap.num_voxels(), vmap.size(), (m_sync - base) / 1e9, (m_reg - base) / 1e9,
                                 os.environ.get("SAGEICP_SIZE_CLASSES", "on")))
    del w, vmap, scan              # (the next workload's baseline must not include this map)
    import gc
    gc.collect()
