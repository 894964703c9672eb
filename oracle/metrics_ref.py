"""numpy restatement of the reference's trajectory metrics — TEST INFRASTRUCTURE ONLY (the
checker of sageicp_metrics_*; imported by tests/ only).

Follows cpp/sage_icp/metrics/Metrics.cpp line by line with numpy in Eigen's place:
  trajectory_distances / last_frame_from_segment_length   :44-70
  sequence errors (lengths 100..800 m, every 10th frame)   :88-136
  seq_error (the 180/3.14 of the reference kept)           :140-155
  absolute_trajectory_error (Eigen::umeyama, no scaling)   :157-191
Parity unpinned by the reference (it holds no vectors); pinned by analytic cases in
tests/test_metrics.py."""
import numpy as np

LENGTHS = [100, 200, 300, 400, 500, 600, 700, 800]


def _dist(poses):
    d = [0.0]
    for i in range(1, len(poses)):
        d.append(d[-1] + float(np.linalg.norm(poses[i - 1][:3, 3] - poses[i][:3, 3])))
    return d


def _last(dist, first, length):
    for i in range(first, len(dist)):
        if dist[i] > dist[first] + length:
            return i
    return -1


def seq_error(gt, res):
    gt, res = np.asarray(gt, float), np.asarray(res, float)
    dist = _dist(gt)
    t_err, r_err, n = 0.0, 0.0, 0
    for first in range(0, len(gt), 10):
        for length in LENGTHS:
            last = _last(dist, first, length)
            if last == -1:
                continue
            d_gt = np.linalg.inv(gt[first]) @ gt[last]
            d_res = np.linalg.inv(res[first]) @ res[last]
            e = np.linalg.inv(d_res) @ d_gt
            d = 0.5 * (e[0, 0] + e[1, 1] + e[2, 2] - 1.0)
            r_err += np.arccos(max(min(d, 1.0), -1.0)) / length
            t_err += np.linalg.norm(e[:3, 3]) / length
            n += 1
    if n == 0:
        return float("nan"), float("nan")
    return np.float32(100.0 * t_err / n), np.float32(100.0 * (r_err / n) / 3.14 * 180.0)


def umeyama_rigid(src, dst):
    """src, dst: (n, 3); the rigid T (4x4) minimising sum |dst - T src|^2 (Umeyama 1991)"""
    ms, md = src.mean(0), dst.mean(0)
    sigma = (dst - md).T @ (src - ms) / len(src)
    U, s, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1
    T = np.eye(4)
    T[:3, :3] = U @ np.diag(S) @ Vt
    T[:3, 3] = md - T[:3, :3] @ ms
    return T


def absolute_trajectory_error(gt, res):
    gt, res = np.asarray(gt, float), np.asarray(res, float)
    A = umeyama_rigid(res[:, :3, 3], gt[:, :3, 3])
    rot, trans = 0.0, 0.0
    for G, R in zip(gt, res):
        E = A @ R
        dR = G[:3, :3] @ E[:3, :3].T
        dt = G[:3, 3] - dR @ E[:3, 3]
        c = np.clip((np.trace(dR) - 1.0) / 2.0, -1.0, 1.0)
        th = np.arccos(c)
        rot += th * th
        trans += float(dt @ dt)
    return np.float32(np.sqrt(rot / len(gt))), np.float32(np.sqrt(trans / len(gt)))
