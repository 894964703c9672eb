"""Independent pure-Python/numpy restatement of the map policy and the semantic NN search, used
only to pin the C++ oracle on small cases (the reference holds no vectors for this path).

Follows cpp/sage_icp/core/VoxelHashMap.hpp:45-70 (AddPoint) and VoxelHashMap.cpp:48-130
(GetCorrespondences) of the reference, written from the source text, not from oracle/.
"""
import math

import numpy as np


def trunc_voxel(p, vs):
    return (int(p[0] / vs), int(p[1] / vs), int(p[2] / vs))   # int() truncates toward zero


class PyMap:
    def __init__(self, voxel_size, max_distance, basic=20, critical=20,
                 basic_labels=(40, 44, 48, 49, 50, 70, 72)):
        self.vs, self.md, self.basic, self.critical = voxel_size, max_distance, basic, critical
        self.basic_labels = set(basic_labels)
        self.vox = {}   # insertion-ordered dict: voxel -> list of points

    def add_points(self, pts):
        for p in np.asarray(pts, dtype=np.float64).reshape(-1, 4):
            key = trunc_voxel(p, self.vs)
            blk = self.vox.get(key)
            if blk is None:
                self.vox[key] = [p.copy()]
                continue
            if len(blk) < self.basic:
                blk.append(p.copy())
                continue
            label = int(p[3])
            if label == 0:
                continue
            if label in self.basic_labels:
                for j, e in enumerate(blk):
                    if int(e[3]) == 0:
                        blk[j] = p.copy()
                        break
            elif len(blk) < self.basic + self.critical:
                blk.append(p.copy())
            else:
                for j, e in enumerate(blk):
                    if int(e[3]) == 0:
                        blk[j] = p.copy()
                        break

    def remove_far(self, origin):
        far = [k for k, b in self.vox.items()
               if float(np.sum((b[0][:3] - origin) ** 2)) > self.md * self.md]
        for k in far:
            del self.vox[k]

    def size(self):
        return sum(len(b) for b in self.vox.values())

    def get_correspondences(self, pts, max_dist, th):
        """returns list of (query index, target point) in query order; None-candidate -> rejected"""
        res = []
        ncand = 0
        for qi, p in enumerate(np.asarray(pts, dtype=np.float64).reshape(-1, 4)):
            kx, ky, kz = trunc_voxel(p, self.vs)
            best, best_d = None, float("inf")
            best_d = np.finfo(np.float64).max
            for i in (kx - 1, kx, kx + 1):
                for j in (ky - 1, ky, ky + 1):
                    for k in (kz - 1, kz, kz + 1):
                        for nb in self.vox.get((i, j, k), ()):
                            ncand += 1
                            dx, dy, dz = nb[0] - p[0], nb[1] - p[1], nb[2] - p[2]
                            d = (dx * dx + dy * dy) + dz * dz     # (v3neighbor - v3point).squaredNorm(): packet reduction
                            if int(nb[3]) == int(p[3]) or int(nb[3] * p[3]) == 0:
                                d = d * th
                            if d < best_d:
                                best, best_d = nb, d
            if best is None:
                continue
            dx, dy, dz = best[0] - p[0], best[1] - p[1], best[2] - p[2]
            # (closest - point).head<3>().norm(): a Block of an expression, scalar reduction 1 + 2 terms
            if math.sqrt(dx * dx + (dy * dy + dz * dz)) < max_dist:
                res.append((qi, best.copy()))
        self.last_candidates = ncand
        return res


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def normal_equations(src, tgt, th):
    """Explicit J^T w J / J^T w r of Registration.cpp:62-90."""
    JTJ = np.zeros((6, 6))
    JTr = np.zeros(6)
    for s, t in zip(src[:, :3], tgt[:, :3]):
        r = s - t
        J = np.hstack([np.eye(3), -hat(s)])
        w = th * th / (th + r @ r) ** 2
        JTJ += J.T @ (w * J)
        JTr += J.T @ (w * r)
    return JTJ, JTr
