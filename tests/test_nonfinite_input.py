"""Non-finite input (SURVEY.md section 8(b) "Errors"): the reference casts coordinates and labels to int
(VoxelHashMap.cpp:52-54,87-88,165) — undefined for NaN / Inf — so every entry that would cast one
refuses the whole call with SAGEICP_ERR_INVALID and changes nothing; where the reference is defined
(the range crop drops a point whose norm is not finite) it is followed.  include/sageicp.h, "NaN/Inf"."""
import os

import numpy as np
import pytest

BAD = [np.nan, np.inf, -np.inf]


def _scene(n_map=4000, n_q=500, seed=5):
    rng = np.random.default_rng(seed)
    mp = rng.uniform(-8, 8, size=(n_map, 4))
    mp[:, 3] = rng.choice([0, 40, 50, 70], size=n_map)
    q = rng.uniform(-8, 8, size=(n_q, 4))
    q[:, 3] = rng.choice([0, 40, 50, 70], size=n_q)
    return mp, q


@pytest.mark.parametrize("bad", BAD)
@pytest.mark.parametrize("col", [0, 2, 3])
def test_add_points_refuses_nonfinite_and_inserts_nothing(sage, bad, col):
    mp, _ = _scene()
    m = sage.VoxelHashMap(1.0, 100.0)
    m.AddPoints(mp[:1000])
    before = m.Pointcloud()
    pts = mp[1000:2000].copy()
    pts[777, col] = bad
    with pytest.raises(sage.SageIcpError) as e:
        m.AddPoints(pts)
    assert e.value.code == sage.ERR_INVALID and "finite" in str(e.value)
    assert np.array_equal(m.Pointcloud(), before)          # not even the 777 good points before it
    m.AddPoints(mp[1000:2000])                              # the map is still usable
    assert m.size() > len(before)


@pytest.mark.parametrize("bad", BAD)
def test_get_correspondences_refuses_nonfinite_queries(sage, bad):
    mp, q = _scene()
    m = sage.VoxelHashMap(1.0, 100.0)
    m.AddPoints(mp)
    q = q.copy()
    q[13, 1] = bad
    with pytest.raises(sage.SageIcpError) as e:
        m.GetCorrespondences(q, 2.0, 0.4)
    assert e.value.code == sage.ERR_INVALID


@pytest.mark.gpu
@pytest.mark.parametrize("loop", [0, 2])
@pytest.mark.parametrize("bad,col", [(np.nan, 0), (np.inf, 1), (-np.inf, 2), (np.nan, 3), (np.inf, 3)])
def test_register_frame_refuses_nonfinite_frames(gpu_sage, oracle, bad, col, loop, monkeypatch):
    monkeypatch.setenv("SAGEICP_LOOP", str(loop))
    mp, q = _scene(20000, 3000)
    m = gpu_sage.VoxelHashMap(1.0, 100.0)
    m.AddPoints(mp)
    good = gpu_sage.register_frame(q, m, gpu_sage.IDENTITY, 2.0, 0.3, 0.4)
    f = q.copy()
    f[1234, col] = bad
    for frame in (f, gpu_sage.Frame(m, f)):               # host buffer and resident frame
        with pytest.raises(gpu_sage.SageIcpError) as e:
            gpu_sage.register_frame(frame, m, gpu_sage.IDENTITY, 2.0, 0.3, 0.4)
        assert e.value.code == gpu_sage.ERR_INVALID and "finite" in str(e.value)
    # the handle is fine afterwards: the same call on the clean frame gives the same pose, bit for bit
    again = gpu_sage.register_frame(q, m, gpu_sage.IDENTITY, 2.0, 0.3, 0.4)
    assert np.array_equal(good, again)


@pytest.mark.gpu
@pytest.mark.parametrize("bad,col", [(np.nan, 1), (np.inf, 0), (np.nan, 3)])
def test_device_update_refuses_nonfinite_and_leaves_the_map(gpu_sage, bad, col):
    mp, q = _scene(20000, 3000)
    m = gpu_sage.VoxelHashMap(1.0, 100.0)
    m.AddPoints(mp)
    m.UpdateOnDevice(q[:1000], gpu_sage.IDENTITY)           # the HBM copy becomes the authority
    before = m.Pointcloud()
    pts = q[1000:2000].copy()
    pts[500, col] = bad
    with pytest.raises(gpu_sage.SageIcpError) as e:
        m.UpdateOnDevice(pts, gpu_sage.IDENTITY)
    assert e.value.code == gpu_sage.ERR_INVALID
    assert np.array_equal(m.Pointcloud(), before)
    m.UpdateOnDevice(q[1000:2000], gpu_sage.IDENTITY)
    assert m.size() >= len(before)


@pytest.mark.gpu
def test_pipeline_drops_nonfinite_coordinates_like_the_reference_and_refuses_nonfinite_labels(gpu_sage):
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(11, 3, points_per_frame=20000)
    a, b = gpu_sage.SageICP(), gpu_sage.SageICP()
    for k, f in enumerate(frames):
        g = np.concatenate([f, f[:5]])                      # five extra points ...
        g[-5:, 0] = [np.nan, np.inf, -np.inf, np.nan, np.inf]     # ... that the range crop drops
        pa, pb = a.RegisterFrame(f)[0], b.RegisterFrame(g)[0]
        assert np.array_equal(pa, pb)
    bad = frames[0].copy()
    bad[100, 3] = np.nan
    with pytest.raises(gpu_sage.SageIcpError) as e:
        a.RegisterFrame(bad)
    assert e.value.code == gpu_sage.ERR_INVALID


def test_add_points_with_a_voxel_index_out_of_range_inserts_nothing(sage):
    """|voxel index| >= 2^20 (a coordinate beyond ~10^6 voxels): refused before anything is taken"""
    mp, _ = _scene()
    m = sage.VoxelHashMap(1.0, 100.0)
    m.AddPoints(mp[:1000])
    before = m.Pointcloud()
    pts = mp[1000:2000].copy()
    pts[900, 0] = 2.0e6
    with pytest.raises(sage.SageIcpError) as e:
        m.AddPoints(pts)
    assert e.value.code == sage.ERR_CAPACITY and "nothing was inserted" in str(e.value)
    assert np.array_equal(m.Pointcloud(), before)
