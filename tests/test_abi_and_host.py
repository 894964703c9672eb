"""CPU-side tests of the product library: the C-ABI shared object loads and exports every symbol
include/sageicp.h declares, the host-side map maintenance matches the oracle, and compute entries
fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import os
import re

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sageicp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sageicp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sage):
    names = _declared_symbols()
    assert len(names) >= 25
    L = ctypes.CDLL(sage.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libsageicp_hip.so does not export %s" % n
    assert sorted(sage.EXPORTED_SYMBOLS) == names, "python binding and header disagree"
    assert L.sageicp_abi_version() == sage.ABI_VERSION == 4


def test_stats_struct_layout_matches_header(sage):
    # 2*i32 + 3*u64 + 5*f64 + 2*u32 + u64 + 64*u32 + u64 + 2*u32   (ABI version 3: same layout as 2)
    assert ctypes.sizeof(sage.Stats) == 8 + 24 + 40 + 8 + 8 + 256 + 8 + 8
    # the struct the C compiler lays out from the header, field by field
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "sageicp.h"\nint main(void){printf("%zu", sizeof(sageicp_stats));' + \
          "".join('printf(" %%zu", offsetof(sageicp_stats, %s));' % f for f, _ in sage.Stats._fields_) + \
          'printf(" %zu %d", sizeof(sageicp_comm_info), SAGEICP_ABI_VERSION);return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert got[0] == ctypes.sizeof(sage.Stats)
    assert got[1:-2] == [getattr(sage.Stats, f).offset for f, _ in sage.Stats._fields_]
    assert got[-2] == ctypes.sizeof(sage.CommInfo) and got[-1] == sage.ABI_VERSION


def test_no_oracle_in_product():
    """The product path must not reference the oracle in any form."""
    pkg = os.path.join(ROOT, "sage-icp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "sage_oracle" not in text and "import oracle" not in text, f
                assert "sgo_" not in text, f


def test_compute_fails_loudly_without_device(sage):
    if sage.device_count() > 0:
        pytest.skip("a HIP device is present")
    m = sage.VoxelHashMap(1.0, 100.0)
    m.AddPoints(np.array([[0.5, 0.5, 0.5, 1.0]]))
    with pytest.raises(sage.SageIcpError) as e:
        m.GetCorrespondences(np.array([[0.5, 0.5, 0.5, 1.0]]), 1.0, 0.4)
    assert e.value.code == sage.ERR_NO_DEVICE
    with pytest.raises(sage.SageIcpError):
        sage.register_frame(np.zeros((3, 4)), m, sage.IDENTITY, 1.0, 0.3, 0.4)
    with pytest.raises(sage.SageIcpError):
        sage.align_clouds(np.zeros((3, 4)), np.zeros((3, 4)), 0.3)
    with pytest.raises(sage.SageIcpError):
        sage.transform_points(sage.IDENTITY, np.zeros((3, 4)))


def test_invalid_arguments_are_rejected(sage):
    with pytest.raises(sage.SageIcpError):
        sage.VoxelHashMap(0.0, 100.0)
    with pytest.raises(sage.SageIcpError):
        sage.VoxelHashMap(1.0, 100.0, 200, 100)      # > 255 points per voxel


def test_empty_map_register_returns_initial_guess(sage):
    # Registration.cpp:119 — needs no device
    m = sage.VoxelHashMap(1.0, 100.0)
    g = np.array([0.1, 0.2, 0.3, 0.9, 1.0, 2.0, 3.0])
    out = sage.register_frame(np.zeros((4, 4)), m, g, 6.0, 0.6, 0.4)
    assert np.array_equal(out, g)


def _sorted(a):
    return a[np.lexsort(a.T)]


@pytest.mark.parametrize("vs,basic,critical", [(1.0, 20, 20), (0.8, 4, 3), (0.3, 1, 1), (2.0, 0, 2)])
def test_host_map_policy_matches_oracle(sage, oracle, vs, basic, critical):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-9, 9, size=(30000, 4))
    pts[:, 3] = rng.choice([0, 0, 40, 44, 50, 70, 71, 80, 10], size=len(pts))
    a = sage.VoxelHashMap(vs, 100.0, basic, critical)
    b = oracle.Map(vs, 100.0, basic, critical)
    for lo in range(0, len(pts), 7000):      # several calls: policy is sequential across calls
        a.AddPoints(pts[lo:lo + 7000])
        b.add_points(pts[lo:lo + 7000])
    assert a.size() == b.size() and a.num_voxels() == b.num_voxels()
    assert np.array_equal(_sorted(a.Pointcloud()), _sorted(b.pointcloud()))


def test_host_map_update_remove_clear_clone(sage, oracle):
    rng = np.random.default_rng(8)
    a = sage.VoxelHashMap(1.0, 15.0)
    b = oracle.Map(1.0, 15.0)
    pose = oracle.se3_exp(np.array([3.0, -2.0, 0.5, 0.02, -0.01, 0.4]))
    for step in range(6):
        pts = rng.uniform(-25, 25, size=(5000, 4))
        pts[:, 3] = rng.choice([0, 40, 50, 80], size=len(pts))
        pose = oracle.se3_mul(pose, oracle.se3_exp(np.array([4.0, 0.3, 0, 0, 0, 0.05])))
        a.Update(pts, pose)
        b.update(pts, pose)
        assert a.size() == b.size() and a.num_voxels() == b.num_voxels()
    pa, pb = _sorted(a.Pointcloud()), _sorted(b.pointcloud())
    assert np.allclose(pa, pb, rtol=0, atol=1e-12)    # pose applied by matrix vs quaternion form
    c = a.clone()
    a.Clear()
    assert a.Empty() and a.size() == 0 and not c.Empty()
    assert np.array_equal(_sorted(c.Pointcloud()), pa)
    a.AddPoints(pts)
    assert not a.Empty()


def test_host_map_update_with_origin(sage, oracle):
    rng = np.random.default_rng(9)
    pts = rng.uniform(-30, 30, size=(8000, 4))
    a = sage.VoxelHashMap(1.0, 12.0)
    b = oracle.Map(1.0, 12.0)
    a.Update(pts, np.array([1.0, 2.0, 3.0]))
    b.add_points(pts)
    b.remove_far(np.array([1.0, 2.0, 3.0]))
    assert a.size() == b.size()
    assert np.array_equal(_sorted(a.Pointcloud()), _sorted(b.pointcloud()))


def test_host_map_survives_heavy_erase_and_reinsert(sage, oracle):
    """backward-shift deletion in the open-addressed table must keep every survivor findable"""
    rng = np.random.default_rng(10)
    a = sage.VoxelHashMap(0.5, 6.0)
    b = oracle.Map(0.5, 6.0)
    for step in range(25):
        c = rng.uniform(-10, 10, size=3)
        pts = rng.normal(size=(2000, 4)) * 4 + np.append(c, 0)
        pts[:, 3] = 40
        a.Update(pts, c)
        b.add_points(pts)
        b.remove_far(c)
        assert a.size() == b.size() and a.num_voxels() == b.num_voxels()
    assert np.array_equal(_sorted(a.Pointcloud()), _sorted(b.pointcloud()))


def test_size_classed_storage_is_invisible_and_smaller(sage, oracle, monkeypatch):
    """Voxels live in regions of 4 / 8 / 16 / max_points_per_voxel points and move up as they fill
    (include/sageicp.h, sageicp_map_point_slots): the same map, point for point and in the same
    Pointcloud() order, as with one full-size region per voxel — in a fraction of the storage."""
    rng = np.random.default_rng(12)
    maps = {}
    for classes in ("1", "0"):
        monkeypatch.setenv("SAGEICP_SIZE_CLASSES", classes)
        maps[classes] = sage.VoxelHashMap(0.5, 9.0)           # the environment is read when the map is created
    monkeypatch.delenv("SAGEICP_SIZE_CLASSES")
    b = oracle.Map(0.5, 9.0)
    for step in range(12):
        c = rng.uniform(-6, 6, size=3)
        # a dense clump (voxels climb through every class up to the 40-point cap) in a sparse cloud
        pts = np.concatenate([rng.normal(size=(3000, 4)) * 0.6, rng.uniform(-8, 8, size=(3000, 4))]) + np.append(c, 0)
        pts[:, 3] = rng.choice([0, 0, 40, 44, 70, 80], size=len(pts))
        for m in maps.values():
            m.Update(pts, c)
        b.add_points(pts)
        b.remove_far(c)
        assert maps["1"].size() == maps["0"].size() == b.size()
        assert maps["1"].num_voxels() == b.num_voxels()
    pc = maps["1"].Pointcloud()
    assert np.array_equal(pc, maps["0"].Pointcloud())            # same order, not only the same set
    assert np.array_equal(_sorted(pc), _sorted(b.pointcloud()))
    assert maps["0"].point_slots() >= 40 * maps["0"].num_voxels()
    assert maps["1"].point_slots() >= maps["1"].size()
    assert maps["1"].point_slots() < 0.5 * maps["0"].point_slots()
    counts = np.unique(np.floor(pc[:, :3] / 0.5), axis=0, return_counts=True)[1]
    assert counts.max() > 16 and (counts <= 4).sum() > 100          # both ends of the class ladder were in use
    # a clone carries the layout along, whatever the environment says by then
    monkeypatch.setenv("SAGEICP_SIZE_CLASSES", "0")
    c1 = maps["1"].clone()
    assert c1.point_slots() == maps["1"].point_slots() and np.array_equal(c1.Pointcloud(), pc)
    # capacities below the class sizes: a single class, rounded up to whole units of 4 points
    monkeypatch.delenv("SAGEICP_SIZE_CLASSES")
    small = sage.VoxelHashMap(1.0, 100.0, 2, 1)
    small.AddPoints(rng.uniform(-5, 5, size=(4000, 4)))
    assert small.point_slots() == 4 * small.num_voxels()


@settings(max_examples=25, deadline=None, derandomize=True)
@given(seed=st.integers(0, 2**31 - 1), vs=st.sampled_from([0.25, 1.0, 3.0]), basic=st.sampled_from([0, 1, 3, 6, 20]),
       critical=st.sampled_from([1, 2, 5, 20, 60]), md=st.sampled_from([4.0, 12.0, 1000.0]),
       n_pts=st.sampled_from([1, 50, 1500]), frames=st.integers(1, 7))
def test_size_classes_property(sage, oracle, seed, vs, basic, critical, md, n_pts, frames):
    """random capacities (below, at and far above the class sizes 4 / 8 / 16), voxel sizes, eviction
    radii and batch sizes: the map with size-classed regions equals the map with one full-size
    region per voxel in content AND order, and the oracle's in content; clones and Clear() included"""
    import os
    rng = np.random.default_rng(seed)
    old = os.environ.get("SAGEICP_SIZE_CLASSES")
    try:
        os.environ["SAGEICP_SIZE_CLASSES"] = "0"
        flat = sage.VoxelHashMap(vs, md, basic, critical)
        os.environ.pop("SAGEICP_SIZE_CLASSES")
        cls = sage.VoxelHashMap(vs, md, basic, critical)
    finally:
        if old is None:
            os.environ.pop("SAGEICP_SIZE_CLASSES", None)
        else:
            os.environ["SAGEICP_SIZE_CLASSES"] = old
    orc = oracle.Map(vs, md, basic, critical)
    for k in range(frames):
        c = rng.uniform(-3 * vs, 3 * vs, size=3)
        pts = np.concatenate([rng.normal(size=(n_pts, 4)) * 0.8 * vs, rng.uniform(-5 * vs, 5 * vs, size=(n_pts, 4))])
        pts[:, :3] += c
        pts[:, 3] = rng.choice([0, 0, 40, 44, 50, 70, 71, 80], size=len(pts))
        if seed % 3 == 0 and k == 2:
            cls, keep = cls.clone(), cls          # go on with the copy; the original must stay intact
            snapshot = keep.Pointcloud()
        for m in (cls, flat):
            m.Update(pts, c)
        orc.add_points(pts)
        orc.remove_far(c)
        assert cls.size() == flat.size() == orc.size() and cls.num_voxels() == flat.num_voxels() == orc.num_voxels()
        if seed % 3 == 0 and k == 2:
            assert np.array_equal(keep.Pointcloud(), snapshot)
        if seed % 5 == 0 and k == 3:
            for m in (cls, flat):
                m.Clear()
            orc = oracle.Map(vs, md, basic, critical)
    a = cls.Pointcloud()
    assert np.array_equal(a, flat.Pointcloud())
    assert np.array_equal(_sorted(a), _sorted(orc.pointcloud()))
    assert cls.point_slots() >= cls.size() and flat.point_slots() >= flat.size()


def test_synthetic_workload_is_deterministic_and_exact(sage):
    from sage_icp_amd import synthetic as syn
    mk = lambda: sage.VoxelHashMap(1.0, 100.0)
    w1 = syn.make_workload("c2", mk, scale=0.02)
    w2 = syn.make_workload("c2", mk, scale=0.02)
    assert w1["map"].size() == 20000 and len(w1["scan"]) == 2400
    assert np.array_equal(w1["scan"], w2["scan"])
    assert np.array_equal(w1["map"].Pointcloud(), w2["map"].Pointcloud())
    r = np.linalg.norm(w1["scan"][:, :3], axis=1)
    assert r.min() > 4.0 and r.max() < 101.0
    assert np.array_equal(w1["scan"][:, :3], w1["scan"][:, :3].astype(np.float32).astype(np.float64))


def test_multi_device_map_replicates_mutations_on_the_host(sage):
    """sageicp_map_set_devices works without a GPU for everything that is host logic: every copy
    of the map takes every mutation, clones keep the device list"""
    rng = np.random.default_rng(9)
    pts = rng.uniform(-20, 20, size=(4000, 4))
    pts[:, 3] = rng.choice([0, 40, 70], size=len(pts))
    a = sage.VoxelHashMap(1.0, 15.0)
    a.set_devices([0, 1, 2])
    assert a.num_devices() == 3
    a.AddPoints(pts)
    b = sage.VoxelHashMap(1.0, 15.0)
    b.AddPoints(pts)
    assert a.size() == b.size() and a.num_voxels() == b.num_voxels()
    a.RemovePointsFarFromLocation([0.0, 0.0, 0.0])
    b.RemovePointsFarFromLocation([0.0, 0.0, 0.0])
    assert np.array_equal(a.Pointcloud(), b.Pointcloud())
    c = a.clone()
    assert c.num_devices() == 3 and c.size() == a.size()
    a.Clear()
    assert a.Empty() and not c.Empty()
    with pytest.raises(sage.SageIcpError):
        a.set_devices(list(range(9)))


def test_round3_entries_on_the_host_side(sage):
    """entries added with ABI version 2 that need no device: which copy of a map is the authority,
    Pointcloud() of a host-authoritative map with a short buffer, what a communicator without an
    RCCL side reports, prefetch cancel, null arguments"""
    import ctypes as C
    L = sage.lib()
    m = sage.VoxelHashMap(1.0, 100.0)
    pts = np.array([[0.5, 0.5, 0.5, 40.0], [0.6, 0.5, 0.5, 40.0], [5.5, 0.5, 0.5, 0.0]])
    m.AddPoints(pts)
    assert not m.resident() and L.sageicp_map_resident(None) == 0
    part = np.full((2, 4), -1.0)
    assert L.sageicp_map_pointcloud(m._h, part.ctypes.data_as(C.POINTER(C.c_double)), 2) == 3
    assert np.array_equal(part, m.Pointcloud()[:2])
    assert L.sageicp_map_pointcloud(m._h, None, 0) == 3 and L.sageicp_map_pointcloud(None, None, 0) == 0
    comm = sage.Comm(None, 1, 4, 0)                       # no RCCL side, nothing connected yet
    info = comm.describe()
    assert (info["rank"], info["nranks"], info["has_rccl"], info["rccl_ranks"], info["rccl_rank"]) == (1, 4, 0, -1, -1)
    assert info["p2p_connected"] == 0 and info["p2p_enabled"] == 0 and info["p2p_poisoned"] == 0
    assert L.sageicp_comm_describe(None, None) == sage.ERR_INVALID
    p = sage.SageICP(sage.make_pipeline_config())
    p.prefetch(np.zeros((5, 4)))
    p.prefetch_cancel()                                    # nothing started yet: a no-op that must not hang
    assert L.sageicp_pipeline_prefetch_cancel(None) == sage.ERR_INVALID


def test_roofline_frac_is_null_without_counters_of_this_build():
    """bench.py: `roofline.frac` is what rocprofv3 measured on exactly this build and loop form, or null — never
    a model under the same name"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h = bench.device_source_hash()
    entry = {"kernel": "k_loop", "source_sha256": h, "hbm_bytes_per_launch": 40_000_000, "avg_launch_us_kernel_trace": 32.0,
             "counters_source": "profiles/rXX/c2-cold/pmc_*.csv.gz (test)", "l2_hit_rate": 0.5}

    def fresh():
        return {"achieved": None, "frac": None, "traffic": None, "counters_fresh": False, "frac_basis": "none"}

    r = bench.apply_counters(fresh(), entry, h, True, 23.3e6, 127.7e6)
    assert r["counters_fresh"] and r["traffic"] == 40_000_000
    assert abs(r["achieved"] - 1250.0) < 1e-6 and abs(r["frac"] - 1250.0 / 8000.0) < 1e-4 and r["l2_hit_rate"] == 0.5
    stale = bench.apply_counters(fresh(), dict(entry, source_sha256="0" * 64), h, True, 23.3e6, 127.7e6)
    assert stale["frac"] is None and stale["achieved"] is None and stale["traffic"] is None and not stale["counters_fresh"]
    assert "another build" in stale["frac_basis"]
    other = bench.apply_counters(fresh(), entry, h, False, 23.3e6, 127.7e6)          # counters of k_loop, the run went through k_icp
    assert other["frac"] is None and not other["counters_fresh"] and "k_loop" in other["frac_basis"]
    assert bench.apply_counters(fresh(), None, h, True, 1.0, 1.0)["frac"] is None
