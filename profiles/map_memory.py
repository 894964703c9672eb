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


for name in (sys.argv[1:] or ["c5"]):
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
