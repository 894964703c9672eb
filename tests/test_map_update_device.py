"""Device-side VoxelHashMap::Update (map_update.hip) against the host map of the
product (bit-identical block contents and order expected) and against the CPU oracle (same set
of points per voxel, same search results).  Needs the MI355X."""
import numpy as np
import os
import pytest


def _examples(n):
    """SAGE_TEST_EXAMPLES=k runs k times the usual number of random examples (campaigns; profiles/README.md)"""
    return n * max(1, int(os.environ.get("SAGE_TEST_EXAMPLES", "1")))
from hypothesis import given, settings, strategies as st

pytestmark = pytest.mark.gpu

LABELS = (0, 0, 40, 44, 50, 70, 71, 80, 99)       # 0 twice: plenty of unlabelled points


def _sorted(a):
    return a[np.lexsort(a.T)]


def _frames(seed, n_frames, n_pts, box, step, labels=LABELS):
    """points in the SENSOR frame + the poses that carry them along a curved path"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_frames):
        p = rng.uniform(-box, box, size=(n_pts, 4))
        p[:, 2] = rng.uniform(-1.5, 2.5, n_pts)
        p[:, 3] = rng.choice(labels, size=n_pts)
        yaw = 0.05 * k
        pose = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), step * k, 0.3 * step * k, 0.0])
        out.append((p, pose))
    return out


def _maps(sage, oracle, vs, md, basic, critical):
    kw = dict(basic_points_per_voxel=basic, critical_points_per_voxel=critical)
    return (sage.VoxelHashMap(vs, md, **kw), sage.VoxelHashMap(vs, md, **kw),
            oracle.Map(vs, md, basic, critical))


@pytest.mark.parametrize("vs,md,basic,critical,n_pts,box,step", [
    (1.0, 25.0, 3, 2, 6000, 9.0, 4.0),       # small capacities: every branch of the policy, evictions
    (0.5, 40.0, 20, 20, 20000, 30.0, 6.0),   # reference capacities, table growth on the device
    (1.0, 12.0, 1, 0, 3000, 10.0, 7.0),      # one point per voxel, heavy eviction / block reuse
    (2.0, 1000.0, 0, 4, 4000, 40.0, 5.0),    # basic = 0: only the first point and critical labels
])
def test_device_update_equals_host_update(gpu_sage, oracle, vs, md, basic, critical, n_pts, box, step):
    sage = gpu_sage
    dev, host, orc = _maps(sage, oracle, vs, md, basic, critical)
    for k, (p, pose) in enumerate(_frames(5, 9, n_pts, box, step)):
        dev.UpdateOnDevice(p, pose)
        host.Update(p, pose)
        w = sage.transform_points(pose, p)
        orc.add_points(w)
        orc.remove_far(pose[4:])
        assert dev.size() == host.size() == orc.size(), "frame %d" % k
        assert dev.num_voxels() == host.num_voxels(), "frame %d" % k
    a, b = dev.Pointcloud(), host.Pointcloud()
    assert np.array_equal(a, b), "blocks differ from the host map (content or order)"
    assert np.array_equal(_sorted(a), _sorted(orc.pointcloud()))


def test_device_update_interleaved_with_host_entries(gpu_sage, oracle):
    """authority moves device -> host -> device; searches in between stay index-exact"""
    sage = gpu_sage
    dev, host, orc = _maps(sage, oracle, 1.0, 30.0, 4, 3)
    rng = np.random.default_rng(3)
    for k, (p, pose) in enumerate(_frames(11, 8, 5000, 10.0, 5.0)):
        if k % 3 == 2:                       # a host-side entry between device updates
            extra = rng.uniform(-8, 8, size=(700, 4)) + np.append(pose[4:], 0)
            extra[:, 3] = rng.choice(LABELS, size=len(extra))
            dev.AddPoints(extra)
            host.AddPoints(extra)
            orc.add_points(extra)
        dev.UpdateOnDevice(p, pose)
        host.Update(p, pose)
        orc.add_points(sage.transform_points(pose, p))
        orc.remove_far(pose[4:])
        q = rng.uniform(-9, 9, size=(1500, 4)) + np.append(pose[4:], 0)
        q[:, 3] = rng.choice(LABELS, size=len(q))
        if k % 2 == 0:                       # GetCorrespondences downloads; RegisterFrame does not
            _, tgt, idx = dev.GetCorrespondences(q, 3.0, 0.4, with_index=True)
            _, otgt, oidx = orc.get_correspondences(q, 3.0, 0.4, with_index=True)
            assert np.array_equal(idx, oidx) and np.array_equal(tgt, otgt), "frame %d" % k
    before = host.Pointcloud()
    p, pose = _frames(12, 1, 4000, 10.0, 5.0)[0]
    dev.UpdateOnDevice(p, pose)             # leaves the HBM copy as the authority ...
    c = dev.clone()                         # ... so this is a device-to-device copy
    host.Update(p, pose)
    assert c.size() == host.size() == dev.size()
    c.UpdateOnDevice(p[:1000], pose)        # the clone lives on its own
    h2 = host.clone()
    h2.Update(p[:1000], pose)
    assert np.array_equal(c.Pointcloud(), h2.Pointcloud())
    assert np.array_equal(dev.Pointcloud(), host.Pointcloud())
    assert not np.array_equal(before, host.Pointcloud())
    dev.Clear()
    assert dev.size() == 0 and dev.Empty()
    assert c.size() == h2.size() > 0


def test_register_frame_on_a_device_updated_map(gpu_sage, oracle):
    """the search runs on the HBM copy a device update left behind (tombstones, claimed slots)"""
    sage = gpu_sage
    dev, host, _ = _maps(sage, oracle, 1.0, 40.0, 20, 20)
    fr = _frames(21, 6, 15000, 25.0, 3.0)
    for p, pose in fr:
        dev.UpdateOnDevice(p, pose)
        host.Update(p, pose)
    scan, pose = fr[-1]
    guess = pose.copy()
    guess[4:] += [0.3, -0.2, 0.05]
    Ta, sa = sage.register_frame(scan[:8000], dev, guess, 3.0, 0.5, 0.4, return_stats=True)
    Tb, sb = sage.register_frame(scan[:8000], host, guess, 3.0, 0.5, 0.4, return_stats=True)
    assert sa.iterations == sb.iterations and sa.n_corr_last == sb.n_corr_last
    assert np.array_equal(Ta, Tb), "same blocks => the same candidates in the same order"


def test_pipeline_with_device_map_update_matches_host_map_update(gpu_sage, oracle):
    sage = gpu_sage
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(9, 6, points_per_frame=30000)
    a = sage.SageICP(sage.make_pipeline_config(map_update_on_device=True))
    b = sage.SageICP(sage.make_pipeline_config(map_update_on_device=False))
    for f in frames:
        pa = a.RegisterFrame(f)[0]
        pb = b.RegisterFrame(f)[0]
        assert np.array_equal(pa, pb)
    assert np.array_equal(a.LocalMap(), b.LocalMap())


def test_device_update_rejects_far_voxel_indices(gpu_sage):
    sage = gpu_sage
    m = sage.VoxelHashMap(0.001, 1e9)
    p = np.array([[0.5, 0.5, 0.5, 40.0], [2000.0, 0.0, 0.0, 40.0]])     # 2e6 voxels away
    with pytest.raises(sage.SageIcpError):
        m.UpdateOnDevice(p, sage.IDENTITY)
    assert m.size() == 0


@settings(max_examples=_examples(12), deadline=None)
@given(seed=st.integers(0, 2**31 - 1), vs=st.sampled_from([0.3, 1.0, 2.5]),
       basic=st.integers(0, 6), critical=st.integers(1, 5), md=st.sampled_from([6.0, 15.0, 400.0]),
       n_pts=st.sampled_from([1, 37, 900, 4000]), frames=st.integers(1, 6))
def test_device_update_property(gpu_sage, vs, basic, critical, md, n_pts, frames, seed):
    """random capacities, voxel sizes, eviction radii and batch sizes: device == host, block for block"""
    sage = gpu_sage
    kw = dict(basic_points_per_voxel=basic, critical_points_per_voxel=critical,
              basic_parts_labels=(40, 44, 50))
    dev, host = sage.VoxelHashMap(vs, md, **kw), sage.VoxelHashMap(vs, md, **kw)
    rng = np.random.default_rng(seed)
    for k, (p, pose) in enumerate(_frames(seed % 1000, frames, n_pts, 6.0 * vs, 2.0 * vs)):
        if seed % 2 and k:
            p[: len(p) // 2, :3] = np.round(p[: len(p) // 2, :3] / vs) * vs     # points on voxel faces
        if (seed >> 1) % 3 == 0 and k == 1:
            extra = rng.uniform(-3 * vs, 3 * vs, size=(50, 4))
            dev.AddPoints(extra)            # a host-side entry in between (download + re-upload)
            host.AddPoints(extra)
        dev.UpdateOnDevice(p, pose)
        host.Update(p, pose)
        assert dev.size() == host.size() and dev.num_voxels() == host.num_voxels()
    assert np.array_equal(dev.Pointcloud(), host.Pointcloud())


def test_pointcloud_of_a_resident_map_keeps_it_resident(gpu_sage, oracle):
    """Pointcloud() while the HBM copy is the authority (the node's LocalMap() every frame,
    OdometryServer.cpp:211-220): packed on the device in block-pool order — byte-identical to the
    host map's — and the map stays where it is: no download of the blocks, no table rebuild, the
    next RegisterFrame / Update find it resident and upload nothing"""
    sage = gpu_sage
    import ctypes as C
    dev, host, _ = _maps(sage, oracle, 1.0, 40.0, 20, 20)
    fr = _frames(31, 7, 15000, 25.0, 3.0)
    scan, pose = fr[-1]
    for k, (p, T) in enumerate(fr):
        dev.UpdateOnDevice(p, T)
        host.Update(p, T)
        assert dev.resident() and not host.resident()
        cloud = dev.Pointcloud()                       # every frame, like the node
        assert dev.resident(), "Pointcloud() must not hand the authority back to the host"
        assert np.array_equal(cloud, host.Pointcloud()), "frame %d" % k
        Ta, sa = sage.register_frame(scan[:6000], dev, T, 3.0, 0.5, 0.4, return_stats=True)
        Tb, sb = sage.register_frame(scan[:6000], host, T, 3.0, 0.5, 0.4, return_stats=True)
        assert dev.resident() and np.array_equal(Ta, Tb) and sa.iterations == sb.iterations
    # a short buffer takes the first `cap` points, the return value is the map's size
    n = dev.size()
    part = np.full((100, 4), -7.0)
    got = sage.lib().sageicp_map_pointcloud(dev._h, part.ctypes.data_as(C.POINTER(C.c_double)), 100)
    assert got == n and np.array_equal(part, host.Pointcloud()[:100])
    assert sage.lib().sageicp_map_pointcloud(dev._h, None, 0) == n and dev.resident()
    # host-side entries still move the authority (and Pointcloud() follows)
    dev.AddPoints(scan[:50] + 1000.0)
    host.AddPoints(scan[:50] + 1000.0)
    assert not dev.resident() and np.array_equal(dev.Pointcloud(), host.Pointcloud())
    dev.Clear()
    assert dev.Pointcloud().shape == (0, 4)


def test_regions_follow_the_counts_and_are_reused(gpu_sage, oracle, monkeypatch):
    """size-classed storage under the device-side update: voxels climb 4 -> 8 -> 16 -> cap as a
    dense clump fills up, evicted and outgrown regions are handed out again, the map stays
    block-for-block the host's (and the map with one full-size class, SAGEICP_SIZE_CLASSES=0) and
    its footprint follows what it holds"""
    sage = gpu_sage
    dev, host, _ = _maps(sage, oracle, 0.5, 14.0, 20, 20)
    monkeypatch.setenv("SAGEICP_SIZE_CLASSES", "0")
    flat = sage.VoxelHashMap(0.5, 14.0, basic_points_per_voxel=20, critical_points_per_voxel=20)
    monkeypatch.delenv("SAGEICP_SIZE_CLASSES")
    rng = np.random.default_rng(44)
    peak = 0
    for k in range(14):
        pose = np.array([0.0, 0.0, 0.0, 1.0, 2.5 * k, 0.0, 0.0])
        # a clump that keeps receiving points (voxels move up a class between and inside passes),
        # a sparse cloud around it (one- and two-point voxels), and long runs into single voxels
        p = np.concatenate([rng.normal(size=(2500, 4)) * 0.7, rng.uniform(-9, 9, size=(2500, 4)),
                            rng.uniform(0.0, 0.45, size=(300, 4)) + np.array([3.0, 3.0, 0.0, 0.0])])
        p[:, 3] = rng.choice(LABELS, size=len(p))
        for m in (dev, flat):
            m.UpdateOnDevice(p, pose)
        host.Update(p, pose)
        assert dev.resident() and dev.size() == host.size() == flat.size()
        assert dev.point_slots() >= dev.size() and flat.point_slots() >= 40 * flat.num_voxels()
        peak = max(peak, dev.point_slots())
    a = dev.Pointcloud()
    assert np.array_equal(a, host.Pointcloud()) and np.array_equal(a, flat.Pointcloud())
    # the high-water mark of the allocator stays near what the map held at its fullest: freed
    # regions are reused, not leaked (the eviction radius keeps about five frames alive)
    assert dev.point_slots() == peak and peak < 0.55 * flat.point_slots()
    later = dev.point_slots()
    for k in range(14, 20):
        pose = np.array([0.0, 0.0, 0.0, 1.0, 2.5 * k, 0.0, 0.0])
        p = rng.uniform(-9, 9, size=(3000, 4))
        p[:, 3] = rng.choice(LABELS, size=len(p))
        dev.UpdateOnDevice(p, pose)
        host.Update(p, pose)
    assert np.array_equal(dev.Pointcloud(), host.Pointcloud())
    assert dev.point_slots() <= later + 4 * 3000          # sparse frames into a map that is shedding its clump
    # authority back to the host (download of table, regions, free stacks) and on
    extra = rng.uniform(-4, 4, size=(500, 4)) + np.array([2.5 * 19, 0, 0, 0])
    dev.AddPoints(extra)
    host.AddPoints(extra)
    assert not dev.resident() and np.array_equal(dev.Pointcloud(), host.Pointcloud())


def test_a_two_million_point_batch_into_an_empty_map(gpu_sage, oracle):
    """the storage reserved for a pass is bounded by what its runs can ask for (a unit per point of
    a new voxel, a last-class region per voxel that can move), not by a last-class region per POINT:
    a first scan of millions of points must not hit the 2^24-unit limit"""
    sage = gpu_sage
    rng = np.random.default_rng(9)
    p = rng.uniform(-90, 90, size=(2_000_000, 4))
    p[:, 2] = rng.uniform(-3, 9, len(p))
    p[:, 3] = rng.choice(LABELS, size=len(p))
    dev = sage.VoxelHashMap(1.0, 400.0)
    host = sage.VoxelHashMap(1.0, 400.0)
    dev.UpdateOnDevice(p, sage.IDENTITY)
    host.Update(p, sage.IDENTITY)
    assert dev.resident() and dev.size() == host.size() and dev.num_voxels() == host.num_voxels()
    assert np.array_equal(dev.Pointcloud(), host.Pointcloud())
    assert dev.size() <= dev.point_slots() < 2.2 * dev.size()
