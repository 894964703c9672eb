"""What deviations D2 / D3 cost on a stream: the 200-frame c3 stream (KITTI file format, scan
correction on) free-running through
  A  the oracle pipeline, arrival-order emission + collect-then-erase
  B  the oracle pipeline with tsl::robin_map v1.0.1 order emulated       (what the reference does)
  B1 / B2  only the emission order / only the erase-while-iterating sweep
  G  the GPU pipeline (reference-order emission, collect-then-erase sweep), G0: arrival order
Per-frame pose delta B vs A (the cost of the deviation), G vs A (the product's parity), and the
trajectory metrics of each against the planted motion."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
import sage_icp_amd as sage
from sage_icp_amd import kitti_io, synthetic as syn

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
frames, truth = syn.make_stream(7, n_frames, points_per_frame=pts)
d = tempfile.mkdtemp()
kitti_io.write_sequence(d, frames)
vel, lab = kitti_io.list_sequence(d)
cfg = sage.make_pipeline_config()


def run(make, robin):
    oracle.set_robin_order(robin)
    try:
        p = make()
        out = []
        for v, l in zip(vel, lab):
            f = kitti_io.load_frame(v, l, correct=True)
            r = p.RegisterFrame(f) if hasattr(p, "RegisterFrame") else p.register_frame(f)
            out.append(np.array(r[0]))
        return np.array(out)
    finally:
        oracle.set_robin_order(False)


A = run(lambda: oracle.Pipeline(cfg), False)
B = run(lambda: oracle.Pipeline(cfg), True)
B1 = run(lambda: oracle.Pipeline(cfg), 1)      # only the emission order
B2 = run(lambda: oracle.Pipeline(cfg), 2)      # only D2: the sweep erases while iterating
G = G0 = None
if sage.device_count():
    G = run(lambda: sage.SageICP(cfg), False)
    sage.set_downsample_order(False)
    G0 = run(lambda: sage.SageICP(cfg), False)
    sage.set_downsample_order(True)


def delta(X, Y):
    dt, dr = [], []
    for a, b in zip(X, Y):
        e = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(a), b))
        dt.append(np.linalg.norm(e[:3])); dr.append(np.linalg.norm(e[3:]))
    return np.array(dt), np.array(dr)


def mats(p7):
    out = np.tile(np.eye(4), (len(p7), 1, 1))
    for i, p in enumerate(p7):
        out[i, :3, :3] = syn.quat_to_mat(np.asarray(p[:4])); out[i, :3, 3] = p[4:]
    return out


first = np.linalg.inv(mats(truth[:1])[0])
gt = np.array([first @ m for m in mats(truth)])
print("%d frames, %d points per raw scan" % (n_frames, pts))
for name, X, Y in (("B vs A (robin_map order vs arrival order, both CPU)", A, B),
                   ("B2 vs A (only the erase-while-iterating sweep)", A, B2)) + \
        ((("G vs B (GPU product, reference-order emission, vs the reference's behaviour)", B, G),
          ("G vs B1 (the same against the oracle mode it mirrors)", B1, G),
          ("G0 vs A (GPU product in arrival order vs the oracle in arrival order)", A, G0)) if G is not None else ()):
    dt, dr = delta(X, Y)
    print("%-82s per-frame pose delta: max %.3e m %.3e rad, mean %.3e m %.3e rad, last frame %.3e m"
          % (name, dt.max(), dr.max(), dt.mean(), dr.mean(), dt[-1]))
    print("   frames above 1e-4 m or rad: %d of %d" % (int(np.sum((dt > 1e-4) | (dr > 1e-4))), len(dt)))
for name, X in (("A", A), ("B", B)) + ((("G", G),) if G is not None else ()):
    t, r = sage.seq_error(gt, mats(X))
    ar, at = sage.absolute_trajectory_error(gt, mats(X))
    print("%s: KITTI relative error %.4f %% / %.4f deg per 100 m; ATE %.4f m %.5f rad" % (name, t, r, at, ar))
