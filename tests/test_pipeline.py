"""Per-frame pipeline counterpart: the host-side logic around the hot path —
range crop, two-level semantic voxel down-sampling, adaptive threshold, constant-velocity guess,
map update — in the product library vs its restatement in the oracle.

Preprocess + VoxelDownsample run on the device (preprocess.hip, f-3), so every test here needs the
GPU: the two kernels against a pure-Python restatement of core/Preprocessing.cpp, frame 0 of the
pipeline (crop, down-sampling, map update) against the oracle pipeline, and a c3-style synthetic
stream with per-frame pose parity, free-running."""
import numpy as np
import pytest


def _pose_err(oracle, A, B):
    e = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(A), B))
    return np.linalg.norm(e[:3]), np.linalg.norm(e[3:])


def _sorted(a):
    return a[np.lexsort(a.T)]


@pytest.mark.gpu
def test_first_frame_preprocessing_and_map_update_match_oracle(gpu_sage, oracle, reference_emission_order):
    sage = gpu_sage
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


@pytest.mark.gpu
def test_downsample_keeps_first_point_per_voxel_per_group(gpu_sage, oracle, reference_emission_order):
    sage = gpu_sage
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
def test_stream_pose_parity_restarted_and_free_running(gpu_sage, oracle, reference_emission_order):
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


@pytest.mark.gpu
def test_stream_with_prefetch_is_bit_identical(gpu_sage, reference_emission_order):
    """sageicp_pipeline_prefetch: the next frame's Preprocess() + Voxelize() under this frame's ICP
    loop.  Same poses to the bit, same source sizes, same map — also when an announced frame is not
    the one registered next (dropped), and with frames of different sizes alternating."""
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(5, 10, points_per_frame=30000)
    frames = [np.ascontiguousarray(f if k % 3 else f[: len(f) // 2], dtype=np.float64)
              for k, f in enumerate(frames)]
    cfg = gpu_sage.make_pipeline_config()
    a, b = gpu_sage.SageICP(cfg), gpu_sage.SageICP(cfg)
    for k, f in enumerate(frames):
        pa, _, _, ns_a, st_a = a.RegisterFrame(f)
        if k + 1 < len(frames):
            # frame 4 announces a frame that never comes: the prepared clouds must be ignored
            b.prefetch(frames[0] if k == 4 else frames[k + 1])
        pb, _, _, ns_b, st_b = b.RegisterFrame(f)
        assert ns_a == ns_b and st_a.iterations == st_b.iterations, "frame %d" % k
        assert np.array_equal(pa, pb), "frame %d" % k
    assert np.array_equal(a.LocalMap(), b.LocalMap())


@pytest.mark.gpu
def test_prefetch_recognises_a_refilled_buffer_and_can_be_cancelled(gpu_sage, reference_emission_order):
    """the announced frame is matched by pointer, size AND content: a buffer refilled with another
    scan after its clouds were prepared is registered as what it now holds (the stale prepared
    clouds are dropped); a cancelled announcement leaves nothing prepared and nothing reading"""
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(6, 6, points_per_frame=20000)
    n = min(len(f) for f in frames)
    frames = [np.ascontiguousarray(f[:n], dtype=np.float64) for f in frames]
    cfg = gpu_sage.make_pipeline_config()
    a, b = gpu_sage.SageICP(cfg), gpu_sage.SageICP(cfg)
    seq = [frames[0], frames[2], frames[3], frames[4]]
    ref = [a.RegisterFrame(f)[0] for f in seq]
    buf = frames[1].copy()
    b.prefetch(buf)                                   # announces the scan `buf` holds now (frame 1)
    assert np.array_equal(b.RegisterFrame(frames[0])[0], ref[0])
    b.prefetch_wait()                                 # the helper has prepared frame 1's clouds and reads `buf` no more
    buf[:] = frames[2]                                # same pointer, same size, another scan
    assert np.array_equal(b.RegisterFrame(buf)[0], ref[1]), "stale prepared clouds were used"
    b.prefetch(frames[4])
    b.prefetch_cancel()                               # announced, then withdrawn
    assert np.array_equal(b.RegisterFrame(frames[3])[0], ref[2])
    b.prefetch(frames[4])                             # and the ordinary use still works afterwards
    b.prefetch_cancel()
    b.prefetch(frames[4])
    assert np.array_equal(b.RegisterFrame(frames[4])[0], ref[3])
    assert np.array_equal(a.LocalMap(), b.LocalMap())


def _py_preprocess(frame, max_range, min_range, label_max_range):
    out = []
    for p in frame:
        norm = float(np.sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]))     # point.head<3>().norm(): packet reduction
        if norm < max_range and norm > min_range:
            out.append([p[0], p[1], p[2], 0.0 if norm > label_max_range else p[3]])
    return np.array(out).reshape(-1, 4)


def _py_voxel_downsample(frame, voxel_labels, voxel_size, scale):
    """core/Preprocessing.cpp:44-84 restated: first point per voxel per group; emitted group by
    group in input order"""
    grids = [dict() for _ in voxel_labels]
    for p in frame:
        label = int(p[3])
        group = next((g for g, ls in enumerate(voxel_labels) if label in ls), -1)
        if group < 0:
            continue
        vs = voxel_size[group] * scale
        key = (int(p[0] / vs), int(p[1] / vs), int(p[2] / vs))
        if key not in grids[group]:
            grids[group][key] = p
    out = [p for g in grids for p in g.values()]
    return np.array(out).reshape(-1, 4)


@pytest.mark.gpu
def test_device_preprocess_matches_python(gpu_sage):
    rng = np.random.default_rng(31)
    f = rng.normal(size=(20000, 4)) * [40, 40, 3, 0]
    f[:, 3] = rng.choice([0, 10, 40, 50, 70, 81, 252], size=len(f))
    f[:4, :3] = [[5.0, 0, 0], [100.0, 0, 0], [0, 50.0, 0], [3.0, 4.0, 0]]     # on the thresholds
    out = gpu_sage.preprocess(f, 100.0, 5.0, 50.0)
    ref = _py_preprocess(f, 100.0, 5.0, 50.0)
    assert np.array_equal(out, ref) and 0 < len(out) < len(f)
    assert gpu_sage.preprocess(np.zeros((0, 4)), 100.0, 5.0, 50.0).shape == (0, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [0.5, 1.5])
def test_device_voxel_downsample_matches_python(gpu_sage, oracle, scale):
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(33, 1, points_per_frame=60000)
    f = frames[0]
    f[::19, 3] = 77                                     # label of no group -> dropped
    f[1::23, :3] *= -1.0                                # negative coordinates: trunc toward zero
    labels, sizes = gpu_sage.KITTI_VOXEL_LABELS, gpu_sage.KITTI_VOXEL_SIZE
    ref = _py_voxel_downsample(f, labels, sizes, scale)
    gpu_sage.set_downsample_order(False)
    try:
        out = gpu_sage.voxel_downsample(f, labels, sizes, scale)
    finally:
        gpu_sage.set_downsample_order(True)
    assert len(ref) > 1000 and out.shape == ref.shape
    assert np.array_equal(out, ref), "arrival order: same survivors in the same (group, input) order"
    # default: the reference's emission order (tsl::robin_map bucket order, group by group)
    out = gpu_sage.voxel_downsample(f, labels, sizes, scale)
    oracle.set_robin_order(1)
    try:
        oref = oracle.voxel_downsample(f, labels, sizes, scale)
    finally:
        oracle.set_robin_order(0)
    assert np.array_equal(out, oref) and not np.array_equal(out, ref)
    assert np.array_equal(_sorted(out), _sorted(ref))
    # every point in one voxel: a single survivor, the first
    one = np.tile([[0.31, 0.32, 0.33, 40.0]], (500, 1)) + np.arange(500)[:, None] * [1e-6, 0, 0, 0]
    assert np.array_equal(gpu_sage.voxel_downsample(one, [[40]], [1.0], 1.0), one[:1])
