"""Generates tests/golden/*.npz — small input/expected-output vectors for the hot path.

The reference (NeSC-IV/sage-icp) holds no golden vectors for this path and cannot be built or
imported in this image (C++ with absent Eigen/Sophus/TBB/tsl), so these vectors are produced by
the repo's own CPU oracle (oracle/sage_oracle.cpp) after it passed tests/test_oracle_kat.py.
They are regression pins for oracle and HIP path alike, NOT reference output (parity unpinned).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import sage_icp_amd  # noqa: E402  (host-side map only; used to cut the synthetic map stream)
from sage_icp_amd import synthetic as syn  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def scene_case(name, seed, vs, basic, critical, th, max_dist, kernel, n_map, n_scan, half):
    rng = np.random.default_rng(seed)
    stream = syn.sample_surfaces(rng, n_map, half)
    T_gt = syn.pose_from_rpy_t([0.2, -0.1, 1.5], [0.4, -0.2, 0.03])
    scan = syn.make_scan(rng, n_scan, half, T_gt, max_range=1.3 * half, min_range=2.0)
    m = oracle.Map(vs, 100.0, basic, critical)
    m.add_points(stream)
    # GetCorrespondences on the scan moved by the ground-truth pose
    q = oracle.transform_points(T_gt, scan)
    src, tgt, idx = m.get_correspondences(q, max_dist, th, nthreads=1, with_index=True)
    T_step, JTJ, JTr = oracle.align_clouds(src, tgt, kernel, nthreads=1)
    pose, st = m.register_frame(scan, oracle.IDENTITY, max_dist, kernel, th, nthreads=1)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        params=np.array([vs, 100.0, basic, critical, th, max_dist, kernel]),
        map_stream=stream, scan=scan, T_gt=T_gt, queries=q,
        corr_idx=idx, corr_tgt=tgt, sum_candidates=np.array([m.last_sum_candidates]),
        align_pose=T_step, align_JTJ=JTJ, align_JTr=JTr,
        reg_pose=pose, reg_iterations=np.array([st.iterations]),
        reg_n_corr=np.array([st.n_corr_first, st.n_corr_last]),
        map_size=np.array([m.size(), m.num_voxels()]))
    print(name, "map", m.size(), "voxels", m.num_voxels(), "corr", len(idx), "iters",
          st.iterations, "pose", np.round(pose, 5))


if __name__ == "__main__":
    scene_case("street_cold", 101, 1.0, 20, 20, 0.4, 6.0, 2.0 / 3.0, 12000, 600, 20.0)
    scene_case("street_steady", 102, 0.8, 20, 20, 0.05, 0.9, 0.1, 12000, 600, 20.0)
    scene_case("street_small_blocks", 103, 0.5, 3, 2, 0.8, 1.5, 0.3, 10000, 500, 16.0)
