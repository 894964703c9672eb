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
