"""Oracle vs the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "street_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    vs, md, basic, critical, th, max_dist, kernel = g["params"]
    m = oracle.Map(vs, md, int(basic), int(critical))
    m.add_points(g["map_stream"])
    assert [m.size(), m.num_voxels()] == list(g["map_size"])
    src, tgt, idx = m.get_correspondences(g["queries"], max_dist, th, nthreads=2, with_index=True)
    assert np.array_equal(idx, g["corr_idx"]) and np.array_equal(tgt, g["corr_tgt"])
    assert m.last_sum_candidates == g["sum_candidates"][0]
    T, JTJ, JTr = oracle.align_clouds(src, tgt, kernel, nthreads=1)
    pose, st = m.register_frame(g["scan"], oracle.IDENTITY, max_dist, kernel, th, nthreads=1)
    assert st.iterations == g["reg_iterations"][0]
    if oracle.SQNORM3_ORDER == 2:       # the vectors are bit pins of the default association
        assert np.array_equal(JTJ, g["align_JTJ"]) and np.array_equal(T, g["align_pose"])
        assert np.array_equal(pose, g["reg_pose"])
    else:                               # x^2 + (y^2 + z^2): 1 ulp in a residual's weight
        assert np.allclose(JTJ, g["align_JTJ"], rtol=1e-12) and np.allclose(T, g["align_pose"], atol=1e-12)
        assert np.allclose(pose, g["reg_pose"], atol=1e-10)
    # multi-threaded summation order differs at the 1e-13 level only
    pose4, st4 = m.register_frame(g["scan"], oracle.IDENTITY, max_dist, kernel, th, nthreads=4)
    assert np.allclose(pose4, pose, atol=1e-9)


def test_c4_full_fixture_is_sane(oracle):
    """tests/golden/c4_full.npz (make_c4_golden.py: the oracle's full registration of the 500k-vs-10M frame, run
    once in the build container): data only, converged, near the planted pose; the GPU test compares against it."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_full.npz"))
    assert set(g.files) >= {"workload", "map_size", "scan_checksum", "T_gt", "steady_pose", "steady_counts", "cold_pose", "cold_counts"}
    assert all(g[k].dtype.kind in "fiu" for k in g.files)                   # numbers, no text
    assert int(g["map_size"][0]) == 10_000_000 and int(g["workload"][2]) == 500_000
    for params in ("steady", "cold"):
        iters, conv, nc_first, nc_last, sum_cq, sum_corr = [int(x) for x in g[params + "_counts"]]
        assert conv == 1 and 0 < iters < 500 and 0 < nc_first <= nc_last <= 500_000 and sum_cq > sum_corr > 0
        e = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(g["T_gt"]), g[params + "_pose"]))
        assert np.linalg.norm(e[:3]) < 0.05 and np.linalg.norm(e[3:]) < 2e-3
