"""Per-frame pipeline counterpart: the host-side logic around the hot path —
range crop, two-level semantic voxel down-sampling, adaptive threshold, constant-velocity guess,
map update — in the product library vs its restatement in the oracle.

CPU part: everything up to the first RegisterFrame against a non-empty map needs no device
(the first frame registers against an empty map and returns the guess), so the preprocessing and
map-update logic is compared on frame 0.  GPU part: a c3-style synthetic stream, per-frame pose
parity with both sides restarted from the same state, and free-running."""
import numpy as np
import pytest


def _pose_err(oracle, A, B):
    e = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(A), B))
    return np.linalg.norm(e[:3]), np.linalg.norm(e[3:])


def _sorted(a):
    return a[np.lexsort(a.T)]


def test_first_frame_preprocessing_and_map_update_match_oracle(sage, oracle):
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(7, 1, points_per_frame=40000)
    f = frames[0].copy()
    f[::17, 3] = 77          # a label that belongs to no voxel group -> dropped by VoxelDownsample
    f = np.vstack([f, [[1.0, 1.0, 0.0, 40], [150.0, 0.0, 0.0, 40]]])   # below min / above max range
    cfg = sage.make_pipeline_config()
    a = sage.SageICP(cfg)
    b = oracle.Pipeline(cfg)
    pa, _, _, ns_a, _ = a.RegisterFrame(f)
    pb, ns_b, sigma, _ = b.register_frame(f)
    assert np.array_equal(pa, sage.IDENTITY) and np.array_equal(pb, pa)   # empty map: the guess
    assert ns_a == ns_b and 0 < ns_a < len(f)
    assert sigma == 2.0
    ma, mb = a.LocalMap(), b.local_map()
    assert len(ma) == len(mb) > 1000
    assert np.array_equal(_sorted(ma), _sorted(mb))
    r = np.linalg.norm(ma[:, :3], axis=1)
    assert r.min() > 5.0 and r.max() < 100.0
    assert not np.any((r > 50.0) & (ma[:, 3] != 0)), "labels beyond label_max_range are zeroed"
    assert 77 not in ma[:, 3]
    assert len(a.poses()) == 1
    a.reinitialize()
    assert len(a.poses()) == 0 and len(a.LocalMap()) == 0


def test_downsample_keeps_first_point_per_voxel_per_group(sage, oracle):
    # two label groups with different voxel sizes; second point in the same voxel is dropped
    cfg = sage.make_pipeline_config(voxel_labels=[[40], [50]], voxel_size=[1.0, 4.0], min_range=0.1)
    pts = np.array([[10.1, 0.1, 0.1, 40], [10.2, 0.1, 0.1, 40], [10.9, 0.1, 0.1, 40],
                    [20.1, 0.1, 0.1, 50], [21.5, 0.1, 0.1, 50], [22.5, 0.1, 0.1, 50],
                    [30.0, 0.0, 0.0, 60]])
    a = sage.SageICP(cfg)
    b = oracle.Pipeline(cfg)
    a.RegisterFrame(pts)
    b.register_frame(pts)
    ma = _sorted(a.LocalMap())
    assert np.array_equal(ma, _sorted(b.local_map()))
    # scale 0.5: group 40 voxel 0.5 m -> 10.1 and 10.9 kept (10.2 shares 10.1's voxel);
    # group 50 voxel 2.0 m -> 20.1 and 22.5 kept (21.5 shares 20.1's voxel); label 60 dropped
    assert list(ma[:, 0]) == [10.1, 10.9, 20.1, 22.5]


@pytest.mark.gpu
def test_stream_pose_parity_restarted_and_free_running(gpu_sage, oracle):
    from sage_icp_amd import synthetic as syn
    frames, truth = syn.make_stream(11, 12, points_per_frame=30000)
    cfg = gpu_sage.make_pipeline_config()
    a = gpu_sage.SageICP(cfg)
    b = oracle.Pipeline(cfg)
    worst = (0.0, 0.0)
    for k, f in enumerate(frames):
        pa, icp_s, tot_s, ns, st = a.RegisterFrame(f)
        pb, ns_b, sigma, ost = b.register_frame(f)
        dt, dr = _pose_err(oracle, pb, pa)
        worst = (max(worst[0], dt), max(worst[1], dr))
        assert ns == ns_b
        assert dt < 1e-4 and dr < 1e-4, "frame %d: %g m %g rad" % (k, dt, dr)   # north-star bar
        assert dt < 1e-6 and dr < 1e-6, "free-running drift should stay at rounding level"
        if k:
            assert st.iterations == ost.iterations, "frame %d" % k
            assert icp_s > 0 and tot_s >= icp_s
    assert len(a.LocalMap()) == len(b.local_map())
    # the odometry tracks the planted motion (relative pose of the last step)
    rel_est = oracle.se3_mul(oracle.se3_inv(a.poses()[-2]), a.poses()[-1])
    rel_true = oracle.se3_mul(oracle.se3_inv(truth[-2]), truth[-1])
    dt, dr = _pose_err(oracle, rel_true, rel_est)
    assert dt < 0.15 and dr < 0.01
