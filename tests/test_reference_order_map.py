"""A product map in reference-order mode (include/sageicp.h sageicp_map_set_reference_order; host_map.hpp,
robin_order.hpp RobinTable) against the oracle following its emulation of tsl::robin_map
(oracle.set_robin_order(3)) and against the independent Python restatement of tests/test_robin_order.py:
what the far-voxel sweep leaves behind (it erases while iterating, VoxelHashMap.cpp:176-184) and the
order Pointcloud() lists the voxels in (:132-142).  Host code only: no GPU."""
import os

import numpy as np
import pytest


def _examples(n):
    """SAGE_TEST_EXAMPLES=k runs k times the usual number of random examples (campaigns; profiles/README.md)"""
    return n * max(1, int(os.environ.get("SAGE_TEST_EXAMPLES", "1")))

from test_robin_order import PyRobin


def _cloud(rng, n, centre, spread=45.0):
    p = np.zeros((n, 4))
    p[:, :3] = centre + rng.uniform(-spread, spread, size=(n, 3)) * [1.0, 1.0, 0.1]
    p[:, 3] = rng.choice([0, 40, 48, 50, 70, 71, 80, 10], size=n)
    return p


@pytest.fixture
def oracle_ref(oracle):
    oracle.set_robin_order(3)
    yield oracle
    oracle.set_robin_order(False)


def test_mode_is_off_by_default_and_needs_an_empty_map(sage):
    m = sage.VoxelHashMap(1.0, 30.0)
    assert m.reference_order() == 0
    m.AddPoints(np.array([[1.0, 2.0, 3.0, 40.0]]))
    with pytest.raises(sage.SageIcpError):
        m.set_reference_order(True)
    m.Clear()
    m.set_reference_order(True)
    assert m.reference_order() == 1
    m.set_reference_order(False)
    assert m.reference_order() == 0


def test_pointcloud_is_in_bucket_order(sage, oracle_ref):
    rng = np.random.default_rng(11)
    pts = _cloud(rng, 20000, np.zeros(3))
    m = sage.VoxelHashMap(1.0, 100.0).set_reference_order(True)
    m.AddPoints(pts)
    om = oracle_ref.Map(1.0, 100.0)
    om.add_points(pts)
    pc, opc = m.Pointcloud(), om.pointcloud()
    assert np.array_equal(pc, opc)
    # ... which is the independent restatement's order of the voxels' first arrivals
    vox = (pts[:, :3] / 1.0).astype(np.int64)
    r, seen = PyRobin(), set()
    for k in map(tuple, vox):
        if k not in seen:
            seen.add(k)
            r.insert(k, k)
    listed = [tuple(v) for v in (pc[:, :3] / 1.0).astype(np.int64)]
    first = [k for i, k in enumerate(listed) if i == 0 or listed[i - 1] != k]
    assert first == r.order()
    # the default map holds the same points, in another order
    d = sage.VoxelHashMap(1.0, 100.0)
    d.AddPoints(pts)
    dpc = d.Pointcloud()
    assert not np.array_equal(dpc, pc) and sorted(map(tuple, dpc)) == sorted(map(tuple, pc))


def test_sweep_leaves_behind_what_the_reference_leaves_behind(sage, oracle_ref):
    rng = np.random.default_rng(5)
    pts = np.zeros((6000, 4))
    pts[:, :3] = rng.uniform(-60, 60, size=(6000, 3))
    pts[:, 3] = 40
    m = sage.VoxelHashMap(1.0, 30.0).set_reference_order(True)
    d = sage.VoxelHashMap(1.0, 30.0)
    om = oracle_ref.Map(1.0, 30.0)
    for x in (m, d):
        x.AddPoints(pts)
    om.add_points(pts)
    origin = np.zeros(3)
    m.RemovePointsFarFromLocation(origin)
    d.RemovePointsFarFromLocation(origin)
    om.remove_far(origin)
    assert m.num_voxels() == om.num_voxels() > d.num_voxels()          # survivors of the first sweep
    assert np.array_equal(m.Pointcloud(), om.pointcloud())
    far = np.linalg.norm(m.Pointcloud()[:, :3], axis=1) > 30.0 + 2.0
    assert far.any()
    m.RemovePointsFarFromLocation(origin)
    om.remove_far(origin)
    assert m.num_voxels() == om.num_voxels()
    assert np.array_equal(m.Pointcloud(), om.pointcloud())


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_moving_sensor_stream_maps_stay_identical(sage, oracle_ref, seed):
    """Update(points, origin) frame after frame while the sensor moves out of the map's range: inserts,
    growth of the bucket array, sweeps with survivors, voxels re-created after eviction"""
    rng = np.random.default_rng(100 + seed)
    m = sage.VoxelHashMap(1.0, 40.0, 5, 5, [40, 48, 50]).set_reference_order(True)
    om = oracle_ref.Map(1.0, 40.0, 5, 5, [40, 48, 50])
    survivors = 0
    for f in range(40):
        centre = np.array([6.0 * f, 2.0 * np.sin(f / 3.0), 0.0])
        if f == 25:
            centre[0] = 30.0                     # a jump back: evicted ground is seen again
        pts = _cloud(rng, 2500, centre)
        m.Update(pts, centre)
        om.add_points(pts)
        om.remove_far(centre)
        assert m.num_voxels() == om.num_voxels() and m.size() == om.size()
        pc = m.Pointcloud()
        assert np.array_equal(pc, om.pointcloud()), "frame %d" % f
        survivors += int((np.linalg.norm(pc[:, :3] - centre, axis=1) > 40.0 + 2.0).any())
    assert survivors > 0, "the stream was meant to leave far voxels behind at least once"
    assert m.reference_order() == 1


def test_clear_keeps_the_bucket_array_and_copies_carry_it(sage, oracle_ref):
    rng = np.random.default_rng(7)
    a, b = _cloud(rng, 8000, np.zeros(3)), _cloud(rng, 300, np.zeros(3))
    m = sage.VoxelHashMap(1.0, 100.0).set_reference_order(True)
    om = oracle_ref.Map(1.0, 100.0)
    m.AddPoints(a); om.add_points(a)
    m.Clear(); om.clear()
    assert m.Empty() and m.reference_order() == 1
    m.AddPoints(b); om.add_points(b)
    assert np.array_equal(m.Pointcloud(), om.pointcloud())
    # 300 points in a table sized for 8,000: not the order of a fresh map
    fresh = sage.VoxelHashMap(1.0, 100.0).set_reference_order(True)
    fresh.AddPoints(b)
    assert not np.array_equal(fresh.Pointcloud(), m.Pointcloud())
    # a copy iterates like the original, and goes its own way afterwards
    c = m.clone()
    assert c.reference_order() == 1 and np.array_equal(c.Pointcloud(), m.Pointcloud())
    extra = _cloud(rng, 500, np.array([10.0, 0.0, 0.0]))
    c.AddPoints(extra)
    om.add_points(extra)
    assert np.array_equal(c.Pointcloud(), om.pointcloud())
    assert m.size() < c.size()


def test_environment_switch_for_unchanged_callers(sage, oracle_ref):
    rng = np.random.default_rng(9)
    pts = _cloud(rng, 3000, np.zeros(3))
    os.environ["SAGEICP_MAP_REFERENCE_ORDER"] = "1"
    try:
        m = sage.VoxelHashMap(1.0, 100.0)
    finally:
        del os.environ["SAGEICP_MAP_REFERENCE_ORDER"]
    assert m.reference_order() == 1
    m.AddPoints(pts)
    om = oracle_ref.Map(1.0, 100.0)
    om.add_points(pts)
    assert np.array_equal(m.Pointcloud(), om.pointcloud())


def test_order_beyond_the_modelled_probe_distance_is_reported(sage):
    """200 voxels with one hash value: probe distances pass 128, the order is no longer claimed (-1), the map
    keeps every point"""
    from test_robin_order import _voxels_with_hash
    keys = _voxels_with_hash(lambda h: h == 0x12345, 200, np.random.default_rng(29), span=3000)
    c = lambda k: k + (0.5 if k >= 0 else -0.5)          # noqa: E731  (the middle of voxel k under truncation)
    pts = np.array([[c(k[0]), c(k[1]), c(k[2]), 40.0] for k in keys])
    m = sage.VoxelHashMap(1.0, 1e6).set_reference_order(True)
    m.AddPoints(pts)
    assert m.num_voxels() == len(keys) == 200
    assert m.reference_order() == -1
    assert sorted(map(tuple, m.Pointcloud())) == sorted(map(tuple, pts))


@pytest.mark.gpu
def test_streamed_pipeline_with_a_reference_order_map(gpu_sage, oracle_ref):
    """the per-frame pipeline over a sensor that leaves its map behind (local map range 30 m, 2.5 m per
    frame): with SAGEICP_MAP_REFERENCE_ORDER=1 the map the node would publish — LocalMap() — is the
    oracle's in mode 3 (bucket order, survivors of the sweeps), point for point IN ORDER, the poses
    agree as always; the default pipeline evicts more"""
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(21, 30, points_per_frame=20000, step=(2.5, 0.0, 0.0), max_range=30.0)
    cfg = gpu_sage.make_pipeline_config(max_range=30.0, local_map_range=30.0)
    os.environ["SAGEICP_MAP_REFERENCE_ORDER"] = "1"
    try:
        a = gpu_sage.SageICP(cfg)
    finally:
        del os.environ["SAGEICP_MAP_REFERENCE_ORDER"]
    d = gpu_sage.SageICP(cfg)
    b = oracle_ref.Pipeline(cfg)
    differs = 0
    for k, f in enumerate(frames):
        pa = a.RegisterFrame(f)[0]
        pd = d.RegisterFrame(f)[0]
        pb = b.register_frame(f)[0]
        e = oracle_ref.se3_log(oracle_ref.se3_mul(oracle_ref.se3_inv(pb), pa))
        assert np.linalg.norm(e) < 1e-6, "frame %d" % k
        ma, mb, md = a.LocalMap(), b.local_map(), d.LocalMap()
        assert ma.shape == mb.shape, "frame %d" % k
        assert np.array_equal(ma[:, 3], mb[:, 3]) and np.allclose(ma[:, :3], mb[:, :3], rtol=0, atol=1e-8), "frame %d" % k
        differs += int(len(md) != len(ma))
    assert differs > 0, "the default sweep was expected to evict voxels the reference's leaves for a later frame"


def test_random_update_sequences_against_the_oracle(sage, oracle_ref):
    """property test on the host: random voxel sizes, ranges, capacities, frame sizes and sensor paths (incl.
    jumps back into evicted ground and Clear() in the middle) — after every Update the product's Pointcloud()
    is the oracle's in mode 3, byte for byte and in order"""
    from hypothesis import given, settings, HealthCheck, strategies as st

    @settings(max_examples=_examples(40), deadline=None, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2**31 - 1), vs=st.sampled_from([0.5, 1.0, 2.0]), rng_m=st.sampled_from([8.0, 20.0, 45.0]),
           basic=st.integers(1, 6), critical=st.integers(0, 6), step=st.sampled_from([0.0, 3.0, 9.0, 25.0]),
           n_pts=st.sampled_from([1, 40, 600, 3000]))
    def run(seed, vs, rng_m, basic, critical, step, n_pts):
        rng = np.random.default_rng(seed)
        m = sage.VoxelHashMap(vs, rng_m, basic, critical, [40, 50]).set_reference_order(True)
        om = oracle_ref.Map(vs, rng_m, basic, critical, [40, 50])
        centre = np.zeros(3)
        for f in range(10):
            if f == 6 and seed % 3 == 0:
                m.Clear(); om.clear()
            centre = centre + [step, rng.normal() * 0.5, 0.0] if seed % 5 or f != 7 else np.zeros(3)
            pts = _cloud(rng, n_pts, centre, spread=rng_m * 1.2)
            m.Update(pts, centre)
            om.add_points(pts)
            om.remove_far(centre)
            assert m.size() == om.size() and m.num_voxels() == om.num_voxels()
            assert np.array_equal(m.Pointcloud(), om.pointcloud())
        assert m.reference_order() == 1

    run()


@pytest.mark.gpu
def test_random_update_sequences_on_the_device_against_the_oracle(gpu_sage, oracle_ref):
    """the same property with Update(points, pose) on the DEVICE (sageicp_map_update_pose_device on a map in
    reference-order mode: the device inserts and finds the far voxels, the host replays only the voxels concerned on
    its bucket array — RobinTable::sweep_erase_listed — and the device evicts what the sweep reached): after every
    update the map is the oracle's in mode 3, Pointcloud() byte for byte and IN ORDER, also across Clear(), jumps back
    into evicted ground, and a switch to host-side updates and back"""
    from hypothesis import example, given, settings, HealthCheck, strategies as st

    # (the example: a 64-bucket table whose run wraps around the end of the array — the listed sweep of round 6's first
    # form left one voxel too many; tests/test_robin_order.py holds the host-side half of it)
    @settings(max_examples=_examples(40), deadline=None, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2**31 - 1), vs=st.sampled_from([0.5, 1.0, 2.0]), rng_m=st.sampled_from([8.0, 20.0, 45.0]),
           basic=st.integers(1, 6), critical=st.integers(0, 6), step=st.sampled_from([0.0, 3.0, 9.0, 25.0]),
           n_pts=st.sampled_from([1, 40, 600, 3000]))
    @example(seed=1356125, vs=2.0, rng_m=45.0, basic=1, critical=0, step=0.0, n_pts=40)
    def run(seed, vs, rng_m, basic, critical, step, n_pts):
        rng = np.random.default_rng(seed)
        m = gpu_sage.VoxelHashMap(vs, rng_m, basic, critical, [40, 50]).set_reference_order(True)
        om = oracle_ref.Map(vs, rng_m, basic, critical, [40, 50])
        centre = np.zeros(3)
        for f in range(10):
            if f == 6 and seed % 3 == 0:
                m.Clear(); om.clear()
            centre = centre + [step, rng.normal() * 0.5, 0.0] if seed % 5 or f != 7 else np.zeros(3)
            pts = _cloud(rng, n_pts, np.zeros(3), spread=rng_m * 1.2)       # in the sensor frame
            pose = np.array([0.0, 0.0, 0.0, 1.0, centre[0], centre[1], centre[2]])
            if f == 4 and seed % 2 == 0:
                m.Update(oracle_ref.transform_points(pose, pts), centre)    # (a host-side update in between)
            else:
                m.UpdateOnDevice(pts, pose)
            om.update(pts, pose)
            assert m.size() == om.size() and m.num_voxels() == om.num_voxels(), "frame %d" % f
            assert np.array_equal(m.Pointcloud(), om.pointcloud()), "frame %d" % f
        assert m.reference_order() == 1

    run()


@pytest.mark.gpu
def test_copy_of_a_device_resident_reference_order_map(gpu_sage, oracle_ref):
    """the copy of a map keeps its mode and its bucket array (OdometryServer.cpp:104 copy-assigns the
    pipeline), also when the points live in HBM: the copy lists the same Pointcloud() and goes on exactly like the original"""
    rng = np.random.default_rng(77)
    m = gpu_sage.VoxelHashMap(1.0, 25.0).set_reference_order(True)
    om = oracle_ref.Map(1.0, 25.0)
    pose = lambda c: np.array([0.0, 0.0, 0.0, 1.0, c, 0.0, 0.0])          # noqa: E731
    for f in range(4):
        pts = _cloud(rng, 4000, np.zeros(3), spread=30.0)
        m.UpdateOnDevice(pts, pose(6.0 * f))
        om.update(pts, pose(6.0 * f))
    c = m.clone()
    assert c.reference_order() == 1 and np.array_equal(c.Pointcloud(), om.pointcloud())
    for f in range(4, 7):
        pts = _cloud(rng, 4000, np.zeros(3), spread=30.0)
        for x in (m, c):
            x.UpdateOnDevice(pts, pose(6.0 * f))
        om.update(pts, pose(6.0 * f))
        assert np.array_equal(c.Pointcloud(), om.pointcloud()) and np.array_equal(m.Pointcloud(), om.pointcloud())
