"""Golden vectors for the KITTI-format I/O either side of the path, produced by the
REFERENCE's own code run in the build container — functions of /root/reference/eval/kitti_pub.py:

    correct_scan      :55-84    HDL-64 scan correction (0.205 deg about p x z)
    read_calib_file   :243-287  calib.txt -> 4x4 matrices
    read_poses_file   :289-312  poses.txt -> LiDAR-frame ground truth  Tr^-1 P Tr
    convertdata       :148-159  .label words -> uint8 semantic ids

The module itself cannot be imported here (its first lines import rclpy / sensor_msgs /
tf_transformations, none of which is in the image), but these four are plain numpy functions: this
script parses the file, compiles exactly those function definitions from its AST as they stand
(with the two names they use from the module's import lines: `np`, `inv` = numpy.linalg.inv,
kitti_pub.py:20,28), runs them on seeded inputs and stores inputs and outputs.  No reference text
is kept: the fixture (kitti_io_ref.npz) is data.

    python tests/golden/make_kitti_io_golden.py        # needs /root/reference
"""
import ast
import os
import tempfile
import numpy as np

REF = "/root/reference/eval/kitti_pub.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kitti_io_ref.npz")
WANTED = ("correct_scan", "read_calib_file", "read_poses_file", "convertdata")


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    defs = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    if sorted(d.name for d in defs) != sorted(WANTED):
        raise SystemExit("not all of %s found in %s" % (WANTED, REF))
    ns = {"np": np, "inv": np.linalg.inv}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return [ns[k] for k in WANTED]


def fmt(v):
    return " ".join("%.17g" % x for x in v)      # round-trips a double exactly


def main():
    correct_scan, read_calib_file, read_poses_file, convertdata = reference_functions()
    rng = np.random.default_rng(20240205)

    # --- scan correction: HDL-64-like returns, float32 values as a .bin holds them
    n = 4096
    r = rng.uniform(2.0, 120.0, n)
    az = rng.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(rng.uniform(-25.0, 3.0, n))
    xyz = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1)
    xyz = xyz.astype(np.float32)
    xyz[:8] = [[10, 0, 0], [0, 10, 0], [-10, 0, 0], [0, -10, 0], [5, 5, -1.7], [50, -20, 0.5],
               [1e-3, 80, -2], [3, 4, 12]]
    out64 = correct_scan(None, xyz.astype(np.float64))    # the float64 call of kitti_pub.py:225-227
    out32 = correct_scan(None, xyz.copy())                 # the float32 call of kitti_pub.py:177-178

    # --- calib.txt / poses.txt: a KITTI-odometry-like calibration and a curved 40-pose trajectory
    def rot(axis, a):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0.0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    Tr = np.eye(4)
    Tr[:3, :3] = rot([1, -1, 1], -2.0 * np.pi / 3.0) @ rot([0.3, 1, 0.2], 0.01)   # ~ velodyne -> camera
    Tr[:3, 3] = [-0.0119, -0.0540, -0.2921]
    calib_rows = {"P0": rng.normal(size=12), "P1": rng.normal(size=12), "P2": rng.normal(size=12),
                  "P3": rng.normal(size=12), "Tr": Tr[:3, :4].reshape(-1)}
    poses_cam = []
    P = np.eye(4)
    for k in range(40):
        step = np.eye(4)
        step[:3, :3] = rot([0.05, 1.0, 0.02], 0.01 + 0.002 * np.sin(0.3 * k))
        step[:3, 3] = [0.01 * np.cos(k), -0.004, 1.1 + 0.1 * np.sin(0.2 * k)]
        P = P @ step
        poses_cam.append(P[:3, :4].reshape(-1).copy())
    poses_cam = np.array(poses_cam)
    with tempfile.TemporaryDirectory() as d:
        cpath, ppath = os.path.join(d, "calib.txt"), os.path.join(d, "poses.txt")
        with open(cpath, "w") as f:
            for k in ("P0", "P1", "P2", "P3", "Tr"):
                f.write("%s: %s\n" % (k, fmt(calib_rows[k])))
        with open(ppath, "w") as f:
            for p in poses_cam:
                f.write(fmt(p) + "\n")
        calib = read_calib_file(cpath)
        poses_lidar = np.array(read_poses_file(ppath, calib))

    # --- labels: instance id in the upper 16 bits, semantic id below (ids <= 251: beyond that the
    # reference's np.array(..., dtype=np.uint8) raises under numpy >= 2, so no vector can be made)
    sem = rng.choice(np.array([0, 1, 10, 11, 13, 15, 18, 20, 30, 31, 32, 40, 44, 48, 49, 50, 51, 52,
                               60, 70, 71, 72, 80, 81, 99, 251]), size=2000)
    raw = ((rng.integers(0, 3000, size=2000).astype(np.int64) << 16) | sem).astype(np.uint32).view(np.int32)
    color = {int(s): (1, 2, 3) for s in np.unique(sem)}
    labels_u8, _ = convertdata(raw.reshape(-1, 1), color)

    np.savez_compressed(
        OUT, xyz=xyz, corrected_f64_input=np.asarray(out64, dtype=np.float64),
        corrected_f32_input=np.asarray(out32, dtype=np.float64),
        calib_keys=np.array(["P0", "P1", "P2", "P3", "Tr"]),
        calib_rows=np.array([calib_rows[k] for k in ("P0", "P1", "P2", "P3", "Tr")]),
        calib_Tr=calib["Tr"], poses_cam=poses_cam, poses_lidar=poses_lidar,
        label_words=raw, labels_u8=labels_u8)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
