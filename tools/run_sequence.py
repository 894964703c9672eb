#!/usr/bin/env python
"""Stream a SemanticKITTI-format sequence (or a synthetic one) through the pipeline counterpart on
the MI355X and write a TUM trajectory — the role of eval/kitti_pub.py + the ROS node, without ROS.

  python tools/run_sequence.py --seq $KITTI_ROOT/sequences/00 --out path.txt
  python tools/run_sequence.py --synthetic 50 --out path.txt

With ground truth (KITTI poses.txt + calib.txt via --gt/--calib, or the synthetic stream's true
poses) it also reports the KITTI relative error and the ATE (metrics/Metrics.cpp:140-191).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic frames")
    ap.add_argument("--out", default="path.txt")
    ap.add_argument("--no-correct", action="store_true", help="skip the KITTI 0.205 deg correction")
    ap.add_argument("--sem-th", type=float, default=0.05)
    ap.add_argument("--gt", help="KITTI poses.txt (camera frame)")
    ap.add_argument("--calib", help="KITTI calib.txt (Tr: velodyne -> camera)")
    args = ap.parse_args()
    import numpy as np
    import sage_icp_amd as sage
    from sage_icp_amd import kitti_io, synthetic

    if args.seq:
        vel, lab = kitti_io.list_sequence(args.seq)
        ts_path = os.path.join(args.seq, "times.txt")
        ts = kitti_io.read_timestamps(ts_path) if os.path.exists(ts_path) else np.arange(len(vel)) * 0.1
        frames = (kitti_io.load_frame(v, lab[i] if lab else None, correct=not args.no_correct)
                  for i, v in enumerate(vel))
        n = len(vel)
    else:
        fr, true_poses = synthetic.make_stream(1, args.synthetic or 20, points_per_frame=120000)
        frames, n, ts = iter(fr), len(fr), np.arange(len(fr)) * 0.1
    pipe = sage.SageICP(sage.make_pipeline_config(sem_th=args.sem_th))
    t0 = time.time()
    icp = 0.0
    # one frame of look-ahead: the file of frame k + 1 is read, and its crop + down-sampling run on
    # the device (pipe.prefetch), while frame k registers
    frames = iter(frames)
    nxt = next(frames, None)
    k = -1
    while nxt is not None:
        f, k = np.ascontiguousarray(nxt, dtype=np.float64), k + 1
        nxt = next(frames, None)
        if nxt is not None:
            nxt = pipe.prefetch(nxt)
        pose, icp_s, tot_s, ns, st = pipe.RegisterFrame(f)
        icp += icp_s
        if k % 10 == 0:
            print("frame %d/%d: %d pts -> %d registered, %d iterations, ICP %.2f ms"
                  % (k, n, len(f), ns, st.iterations, 1e3 * icp_s), file=sys.stderr)
    kitti_io.write_tum(args.out, ts[:n], pipe.poses())
    print("%d frames in %.1f s (ICP %.2f s) -> %s" % (n, time.time() - t0, icp, args.out))

    def mats(p7):
        out = np.tile(np.eye(4), (len(p7), 1, 1))
        for i, p in enumerate(p7):
            out[i, :3, :3] = synthetic.quat_to_mat(p[:4])
            out[i, :3, 3] = p[4:]
        return out

    gt = None
    if args.seq and args.gt and args.calib:
        gt = np.array(kitti_io.read_poses_file(args.gt, kitti_io.read_calib_tr(args.calib)))[:n]   # (n, 4, 4)
    elif not args.seq:
        first = np.linalg.inv(mats(true_poses[:1])[0])
        gt = np.array([first @ m for m in mats(true_poses)])               # relative to frame 0
    if gt is not None:
        est = mats(pipe.poses())
        t_err, r_err = sage.seq_error(gt, est)
        ate_r, ate_t = sage.absolute_trajectory_error(gt, est)
        print("KITTI relative error: %.4f %% translation, %.4f deg/100m rotation; ATE %.4f m, %.5f rad"
              % (t_err, r_err, ate_t, ate_r))


if __name__ == "__main__":
    main()
