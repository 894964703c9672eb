"""Soak of the device-side map update against the host map: random capacities, voxel sizes,
eviction radii, batch sizes, host-side entries in between, size classes on and off — Pointcloud()
equal byte for byte after every sequence.  `python profiles/update_soak.py [sequences]`"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import sage_icp_amd as sage

LABELS = (0, 0, 40, 44, 50, 70, 71, 80, 99)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(20240930)
bad = 0
for t in range(N):
    vs = float(rng.choice([0.2, 0.5, 1.0, 2.5]))
    basic = int(rng.choice([0, 1, 2, 4, 6, 20, 30]))
    critical = int(rng.choice([1, 2, 3, 5, 20, 100]))
    md = float(rng.choice([5.0, 15.0, 60.0, 1000.0]))
    n_pts = int(rng.choice([1, 7, 300, 3000, 20000]))
    frames = int(rng.integers(1, 9))
    os.environ["SAGEICP_SIZE_CLASSES"] = "0" if t % 4 == 3 else "1"
    kw = dict(basic_points_per_voxel=basic, critical_points_per_voxel=critical, basic_parts_labels=(40, 44, 50))
    dev, host = sage.VoxelHashMap(vs, md, **kw), sage.VoxelHashMap(vs, md, **kw)
    for k in range(frames):
        box = float(rng.choice([2.0, 6.0, 20.0])) * vs
        p = rng.uniform(-box, box, size=(n_pts, 4))
        if rng.random() < 0.4:                      # a dense clump: voxels climb through the classes
            p[: n_pts // 2] = rng.normal(size=(n_pts // 2, 4)) * 0.6 * vs
        if rng.random() < 0.3:
            p[: n_pts // 3, :3] = np.round(p[: n_pts // 3, :3] / vs) * vs      # points on voxel faces
        p[:, 3] = rng.choice(LABELS, size=n_pts)
        yaw = 0.07 * k
        pose = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), 1.5 * vs * k, 0.4 * vs * k, 0.0])
        if rng.random() < 0.15:
            extra = rng.uniform(-3 * vs, 3 * vs, size=(40, 4))
            dev.AddPoints(extra)
            host.AddPoints(extra)
        if rng.random() < 0.1:
            dev = dev.clone()
        dev.UpdateOnDevice(p, pose)
        host.Update(p, pose)
        if dev.size() != host.size() or dev.num_voxels() != host.num_voxels():
            break
    ok = dev.size() == host.size() and np.array_equal(dev.Pointcloud(), host.Pointcloud())
    if not ok:
        bad += 1
        print("MISMATCH in sequence", t, dict(vs=vs, basic=basic, critical=critical, md=md, n_pts=n_pts, frames=frames))
print("%d sequences, %d mismatches" % (N, bad))
