"""KITTI-format I/O either side of the hot path: what the reference's dataset
publisher eval/kitti_pub.py reads and writes, as plain numpy (no ROS, no kiss_icp, no pykitti).

  read_velodyne / read_labels / load_frame   eval/kitti_pub.py:148-210  (.bin float32 Nx4, .label
                                              int32 & 0xFFFF -> uint8, KITTI 0.205 deg correction,
                                              float32 coordinates widened by the ROS node,
                                              ros/ros2/Utils.hpp:161-180)
  correct_kitti_scan                          eval/kitti_pub.py:49-52 (kiss_icp_pybind._correct_kitti_scan;
                                              the math is spelled out at :55-84)
  read_timestamps                             eval/kitti_pub.py:109-135 (times.txt, 0.0 -> 0.0001)
  read_calib_tr / read_poses_file             eval/kitti_pub.py:266-312 (GT pose -> LiDAR frame,
                                              Tr^-1 P Tr)
  write_tum                                   eval/kitti_pub.py:442-447 (t x y z qx qy qz qw)
Directory layout (SemanticKITTI): <seq>/velodyne/*.bin, <seq>/labels/*.label, <seq>/times.txt,
<seq>/calib.txt, <seq>/poses.txt.
"""
import glob
import os

import numpy as np

VERTICAL_ANGLE_OFFSET = 0.205 * np.pi / 180.0


def correct_kitti_scan(xyz):
    """Rotate every point by 0.205 deg about (p x z): KISS-ICP's HDL-64 intrinsic correction.
    Computed in float64, as kiss_icp_pybind does on a Vector3dVector."""
    p = np.asarray(xyz, dtype=np.float64)
    axis = np.cross(p, np.array([0.0, 0.0, 1.0]))
    nrm = np.linalg.norm(axis, axis=1, keepdims=True)
    ok = nrm[:, 0] > 0
    axis[ok] /= nrm[ok]
    c, s = np.cos(VERTICAL_ANGLE_OFFSET), np.sin(VERTICAL_ANGLE_OFFSET)
    # Rodrigues: p c + (a x p) s + a (a.p)(1 - c); a is perpendicular to p, so the last term is 0
    out = p * c + np.cross(axis, p) * s + axis * np.sum(axis * p, axis=1, keepdims=True) * (1 - c)
    out[~ok] = p[~ok]          # points on the z axis have no rotation axis
    return out


def read_velodyne(path):
    """.bin -> (N, 3) float32 xyz (the intensity column is dropped, kitti_pub.py:176-177)"""
    scan = np.fromfile(path, dtype=np.float32).reshape(-1, 4)
    return scan[:, :3].astype(np.float32)


def read_labels(path):
    """.label -> (N,) uint8 semantic ids.  kitti_pub.py:153,159: `& 0xFFFF`, then np.uint8: ids
    252..259 (moving classes) wrap modulo 256 under numpy < 2 (numpy >= 2 raises); the wrap is
    applied explicitly here."""
    raw = np.fromfile(path, dtype=np.int32)
    return ((raw & 0xFFFF) & 0xFF).astype(np.uint8)


def load_frame(bin_path, label_path=None, correct=True):
    """One scan as the ROS node hands it to the pipeline: (N, 4) float64 (x, y, z, label) whose
    coordinates are float32 values (publisher casts to f32 at kitti_pub.py:178, Utils.hpp widens)."""
    xyz = read_velodyne(bin_path)
    if correct:
        xyz = correct_kitti_scan(xyz).astype(np.float32)
    out = np.zeros((len(xyz), 4), dtype=np.float64)
    out[:, :3] = xyz
    if label_path is not None:
        lab = read_labels(label_path)
        if len(lab) != len(xyz):
            raise ValueError("label count %d != point count %d" % (len(lab), len(xyz)))
        out[:, 3] = lab
    return out


def list_sequence(seq_dir):
    vel = sorted(glob.glob(os.path.join(seq_dir, "velodyne", "*.bin")))
    lab = sorted(glob.glob(os.path.join(seq_dir, "labels", "*.label")))
    if lab and len(lab) != len(vel):
        raise ValueError("%d scans but %d label files" % (len(vel), len(lab)))
    return vel, lab


def read_timestamps(path):
    ts = []
    with open(path) as f:
        for line in f:
            if line.strip():
                v = float(line)
                ts.append(0.0001 if v == 0.0 else v)
    return np.array(ts)


def read_calib_tr(path):
    """calib.txt -> 4x4 `Tr` (velodyne -> camera)"""
    with open(path) as f:
        for line in f:
            key, _, content = line.partition(":")
            if key.strip() == "Tr":
                v = [float(x) for x in content.split()]
                T = np.eye(4)
                T[:3, :4] = np.array(v).reshape(3, 4)
                return T
    raise ValueError("no Tr entry in " + path)


def read_poses_file(path, Tr):
    """poses.txt (camera frame, 12 values per line) -> list of 4x4 LiDAR-frame poses Tr^-1 P Tr"""
    Tr_inv = np.linalg.inv(Tr)
    out = []
    with open(path) as f:
        for line in f:
            v = [float(x) for x in line.split()]
            if len(v) != 12:
                continue
            P = np.eye(4)
            P[:3, :4] = np.array(v).reshape(3, 4)
            out.append(Tr_inv @ P @ Tr)
    return out


def write_tum(path, timestamps, poses7):
    """TUM trajectory: `t x y z qx qy qz qw` per line; poses7 rows are (qx,qy,qz,qw,tx,ty,tz)."""
    with open(path, "w") as f:
        for t, p in zip(timestamps, poses7):
            f.write("%.6f %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n"
                    % (t, p[4], p[5], p[6], p[0], p[1], p[2], p[3]))


def read_tum(path):
    a = np.loadtxt(path).reshape(-1, 8)
    return a[:, 0], np.c_[a[:, 4:8], a[:, 1:4]]


def write_sequence(seq_dir, frames, timestamps=None):
    """Write scans (N,4) [x,y,z,label] as a SemanticKITTI-format sequence (tests / synthetic
    streams): coordinates as float32, intensity 0, labels in the low 16 bits."""
    os.makedirs(os.path.join(seq_dir, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(seq_dir, "labels"), exist_ok=True)
    for i, f in enumerate(frames):
        scan = np.zeros((len(f), 4), dtype=np.float32)
        scan[:, :3] = f[:, :3]
        scan.tofile(os.path.join(seq_dir, "velodyne", "%06d.bin" % i))
        f[:, 3].astype(np.int32).tofile(os.path.join(seq_dir, "labels", "%06d.label" % i))
    ts = np.arange(len(frames)) * 0.1 if timestamps is None else timestamps
    with open(os.path.join(seq_dir, "times.txt"), "w") as fh:
        for t in ts:
            fh.write("%e\n" % t)
