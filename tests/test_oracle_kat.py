"""Known-answer tests that pin the CPU oracle (oracle/sage_oracle.cpp).

The reference ships no tests or vectors for this path, so the oracle is
pinned against independent implementations: scipy Rotation / matrix exponentials for SE3,
numpy.linalg for the 6x6 solve, a pure-Python restatement of the map policy and the semantic NN
(tests/pyref.py), and analytic identities."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation

import pyref


def T_to_mat(T):
    M = np.eye(4)
    M[:3, :3] = Rotation.from_quat(T[:4]).as_matrix()
    M[:3, 3] = T[4:]
    return M


def twist_mat(x):
    M = np.zeros((4, 4))
    M[:3, :3] = pyref.hat(x[3:])
    M[:3, 3] = x[:3]
    return M


# ---------------------------------------------------------------- KAT-1 SE3
@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1e-3, 0.3, 2.5])
def test_se3_exp_matches_matrix_exponential(oracle, scale):
    rng = np.random.default_rng(1)
    for _ in range(20):
        x = rng.normal(size=6) * scale
        T = oracle.se3_exp(x)
        assert np.allclose(T_to_mat(T), expm(twist_mat(x)), atol=1e-12)
        assert abs(np.linalg.norm(T[:4]) - 1) < 1e-14


@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1e-3, 0.3, 1.5])
def test_se3_log_inverts_exp(oracle, scale):
    rng = np.random.default_rng(2)
    for _ in range(20):
        x = rng.normal(size=6) * scale
        if np.linalg.norm(x[3:]) > 3.0:      # log is the principal branch: |omega| < pi
            x[3:] *= 3.0 / np.linalg.norm(x[3:])
        back = oracle.se3_log(oracle.se3_exp(x))
        assert np.allclose(back, x, rtol=1e-9, atol=1e-15)


def test_se3_log_matches_matrix_log(oracle):
    rng = np.random.default_rng(3)
    for _ in range(10):
        x = rng.normal(size=6) * 0.7
        T = oracle.se3_exp(x)
        L = np.real(logm(T_to_mat(T)))
        got = oracle.se3_log(T)
        assert np.allclose(got[:3], L[:3, 3], atol=1e-10)
        assert np.allclose(pyref.hat(got[3:]), L[:3, :3], atol=1e-10)


def test_se3_group_identities(oracle):
    rng = np.random.default_rng(4)
    A = oracle.se3_exp(rng.normal(size=6))
    B = oracle.se3_exp(rng.normal(size=6))
    AB = oracle.se3_mul(A, B)
    assert np.allclose(T_to_mat(AB), T_to_mat(A) @ T_to_mat(B), atol=1e-13)
    I = oracle.se3_mul(A, oracle.se3_inv(A))
    assert np.allclose(T_to_mat(I), np.eye(4), atol=1e-13)
    p = rng.normal(size=3) * 50
    assert np.allclose(oracle.se3_apply(A, p), T_to_mat(A)[:3, :3] @ p + A[4:], atol=1e-12)


def test_transform_points_keeps_label(oracle):
    rng = np.random.default_rng(5)
    T = oracle.se3_exp(rng.normal(size=6) * 0.2)
    pts = rng.normal(size=(100, 4)) * 30
    pts[:, 3] = rng.integers(0, 100, 100)
    out = oracle.transform_points(T, pts)
    assert np.array_equal(out[:, 3], pts[:, 3])
    assert np.allclose(out[:, :3], pts[:, :3] @ T_to_mat(T)[:3, :3].T + T[4:], atol=1e-12)


# ---------------------------------------------------------------- KAT-2 6x6 solve
def test_ldlt_spd_matches_numpy(oracle):
    rng = np.random.default_rng(6)
    for _ in range(50):
        M = rng.normal(size=(6, 6))
        A = M @ M.T + 1e-3 * np.eye(6)
        b = rng.normal(size=6)
        assert np.allclose(oracle.ldlt_solve6(A, b), np.linalg.solve(A, b), rtol=1e-8, atol=1e-10)


def test_ldlt_badly_scaled_needs_pivoting(oracle):
    # ICP-like scaling: translation block ~N, rotation block ~N*|s|^2
    rng = np.random.default_rng(7)
    M = rng.normal(size=(6, 6))
    D = np.diag([1, 1, 1, 1e4, 1e4, 1e4])
    A = D @ (M @ M.T + np.eye(6)) @ D
    b = rng.normal(size=6)
    assert np.allclose(A @ oracle.ldlt_solve6(A, b), b, rtol=1e-7, atol=1e-7)


def test_ldlt_zero_matrix_gives_zero_step(oracle):
    assert np.array_equal(oracle.ldlt_solve6(np.zeros((6, 6)), np.ones(6)), np.zeros(6))


def test_ldlt_rank_deficient_is_finite(oracle):
    # only translation observable (e.g. all correspondences at the origin)
    A = np.zeros((6, 6))
    A[:3, :3] = 5 * np.eye(3)
    b = np.array([1.0, 2.0, 3.0, 0, 0, 0])
    x = oracle.ldlt_solve6(A, b)
    assert np.allclose(x, [0.2, 0.4, 0.6, 0, 0, 0])


# ---------------------------------------------------------------- KAT-3 AlignClouds
def test_align_clouds_normal_equations_match_explicit(oracle):
    rng = np.random.default_rng(8)
    src = rng.normal(size=(200, 4)) * 20
    tgt = src + rng.normal(size=(200, 4)) * 0.1
    T, JTJ, JTr = oracle.align_clouds(src, tgt, 0.5)
    eJ, er = pyref.normal_equations(src, tgt, 0.5)
    assert np.allclose(JTJ, eJ, rtol=1e-12, atol=1e-9)
    assert np.allclose(JTr, er, rtol=1e-12, atol=1e-9)
    x = np.linalg.solve(eJ, -er)
    assert np.allclose(oracle.se3_log(T), x, rtol=1e-7, atol=1e-10)


def test_align_clouds_thread_count_invariant(oracle):
    rng = np.random.default_rng(9)
    src = rng.normal(size=(1000, 4)) * 20
    tgt = src + rng.normal(size=(1000, 4)) * 0.1
    T1, J1, r1 = oracle.align_clouds(src, tgt, 0.3, nthreads=1)
    T4, J4, r4 = oracle.align_clouds(src, tgt, 0.3, nthreads=4)
    assert np.allclose(J1, J4, rtol=1e-12)
    assert np.allclose(T1, T4, atol=1e-13)


def test_align_clouds_recovers_small_motion(oracle):
    rng = np.random.default_rng(10)
    x = np.array([0.01, -0.02, 0.005, 0.001, -0.002, 0.003])
    Tgt = oracle.se3_exp(x)
    src = rng.normal(size=(500, 4)) * 10
    tgt = oracle.transform_points(Tgt, src)
    T, _, _ = oracle.align_clouds(src, tgt, 1.0)
    assert np.allclose(oracle.se3_log(T), x, atol=5e-4)   # one GN step: error O(|x|^2)


def test_align_clouds_no_pairs_is_identity(oracle):
    T, JTJ, JTr = oracle.align_clouds(np.zeros((0, 4)), np.zeros((0, 4)), 1.0)
    assert np.array_equal(T, oracle.IDENTITY)


# ---------------------------------------------------------------- KAT-5 AddPoint policy
def _mk(label, x=0.5):
    return [x, 0.5, 0.5, float(label)]


def test_add_point_policy_table(oracle):
    m = oracle.Map(1.0, 100.0, basic=3, critical=2, basic_labels=(40,))
    # fill the basic part with anything (labels 0, 0, 71)
    m.add_points([_mk(0, 0.1), _mk(0, 0.2), _mk(71, 0.3)])
    assert m.size() == 3
    m.add_points([_mk(0, 0.4)])          # unlabelled after basic part is full -> dropped
    assert m.size() == 3
    m.add_points([_mk(40, 0.5)])         # basic label -> replaces the FIRST label-0 point
    pc = m.pointcloud()
    assert m.size() == 3 and sorted(pc[:, 0]) == [0.2, 0.3, 0.5]
    m.add_points([_mk(80, 0.6), _mk(81, 0.7)])   # critical labels append up to basic+critical
    assert m.size() == 5
    m.add_points([_mk(80, 0.8)])         # full: critical label replaces first label-0 (x=0.2)
    pc = m.pointcloud()
    assert m.size() == 5 and 0.2 not in pc[:, 0] and 0.8 in pc[:, 0]
    m.add_points([_mk(80, 0.9), _mk(40, 0.95)])  # no label-0 left: both dropped
    assert m.size() == 5 and 0.9 not in m.pointcloud()[:, 0]


def test_first_point_of_new_voxel_is_always_kept(oracle):
    m = oracle.Map(1.0, 100.0, basic=0, critical=1, basic_labels=())
    m.add_points([_mk(0)])               # bypasses AddPoint (VoxelHashMap.cpp:171)
    assert m.size() == 1


def test_voxel_index_truncates_toward_zero(oracle):
    m = oracle.Map(1.0, 100.0)
    m.add_points([[-0.5, 0.2, 0.2, 1], [0.5, 0.2, 0.2, 1], [-1.5, 0.2, 0.2, 1]])
    assert m.num_voxels() == 2           # (-0.5) and (0.5) share voxel 0: it is double width


def test_map_matches_python_restatement(oracle):
    rng = np.random.default_rng(11)
    pts = rng.uniform(-6, 6, size=(6000, 4))
    pts[:, 3] = rng.choice([0, 0, 40, 44, 50, 70, 71, 80, 10], size=len(pts))
    m = oracle.Map(1.0, 100.0, basic=4, critical=3)
    pm = pyref.PyMap(1.0, 100.0, basic=4, critical=3)
    m.add_points(pts)
    pm.add_points(pts)
    assert m.size() == pm.size() and m.num_voxels() == len(pm.vox)
    a = m.pointcloud()
    b = np.array([p for blk in pm.vox.values() for p in blk])
    assert np.array_equal(a[np.lexsort(a.T)], b[np.lexsort(b.T)])


def test_remove_far_uses_first_point(oracle):
    m = oracle.Map(1.0, 10.0)
    pm = pyref.PyMap(1.0, 10.0)
    rng = np.random.default_rng(12)
    pts = rng.uniform(-20, 20, size=(3000, 4))
    pts[:, 3] = 40
    for x in (m, pm):
        x.add_points(pts)
        x.remove_far(np.array([1.0, 2.0, 0.0]))
    assert m.size() == pm.size() and 0 < m.size() < 3000


# ---------------------------------------------------------------- KAT-4 semantic NN
def _random_scene(seed, n_map=4000, n_q=600, span=8.0):
    rng = np.random.default_rng(seed)
    mp = rng.uniform(-span, span, size=(n_map, 4))
    mp[:, 3] = rng.choice([0, 40, 50, 70, 71, 80], size=n_map)
    q = rng.uniform(-span - 1, span + 1, size=(n_q, 4))
    q[:, 3] = rng.choice([0, 40, 50, 70, 71, 80], size=n_q)
    return mp, q


@pytest.mark.parametrize("seed,vs,th,md", [(20, 1.0, 0.4, 6.0), (21, 0.8, 0.05, 0.9),
                                            (22, 1.3, 1.0, 2.0), (23, 0.5, 0.4, 0.3)])
def test_get_correspondences_matches_python_restatement(oracle, seed, vs, th, md):
    mp, q = _random_scene(seed)
    m = oracle.Map(vs, 100.0, basic=5, critical=4)
    pm = pyref.PyMap(vs, 100.0, basic=5, critical=4)
    m.add_points(mp)
    pm.add_points(mp)
    src, tgt, idx = m.get_correspondences(q, md, th, nthreads=3, with_index=True)
    ref = pm.get_correspondences(q, md, th)
    assert [r[0] for r in ref] == list(idx)
    assert np.array_equal(tgt, np.array([r[1] for r in ref]).reshape(-1, 4))
    assert np.array_equal(src, q[idx])
    assert m.last_sum_candidates == pm.last_candidates


def test_semantic_scaling_prefers_same_label(oracle):
    m = oracle.Map(1.0, 100.0)
    m.add_points([[0.30, 0.5, 0.5, 50], [0.62, 0.5, 0.5, 40]])
    q = [[0.5, 0.5, 0.5, 40]]
    # nearer point has another label (d=0.2); same-label point at d=0.12... use distinct distances
    m2 = oracle.Map(1.0, 100.0)
    m2.add_points([[0.40, 0.5, 0.5, 50], [0.65, 0.5, 0.5, 40]])   # d = 0.10 (other) vs 0.15 (same)
    _, tgt = m2.get_correspondences(q, 6.0, 0.4)   # 0.15^2*0.4 = 0.009 < 0.10^2 = 0.01
    assert tgt[0, 3] == 40
    _, tgt = m2.get_correspondences(q, 6.0, 1.0)   # semantics off -> geometric nearest
    assert tgt[0, 3] == 50


def test_unlabelled_gets_the_bonus_on_either_side(oracle):
    m = oracle.Map(1.0, 100.0)
    m.add_points([[0.40, 0.5, 0.5, 50], [0.65, 0.5, 0.5, 0]])
    _, tgt = m.get_correspondences([[0.5, 0.5, 0.5, 40]], 6.0, 0.4)
    assert tgt[0, 3] == 0                                  # neighbour unlabelled
    m = oracle.Map(1.0, 100.0)
    m.add_points([[0.40, 0.5, 0.5, 50], [0.65, 0.5, 0.5, 70]])
    _, tgt = m.get_correspondences([[0.5, 0.5, 0.5, 0]], 6.0, 0.4)
    assert tgt[0, 0] == 0.40                               # query unlabelled: both scaled


def test_acceptance_uses_unscaled_distance(oracle):
    m = oracle.Map(1.0, 100.0)
    m.add_points([[0.9, 0.5, 0.5, 40]])
    src, _ = m.get_correspondences([[0.1, 0.5, 0.5, 40]], 0.7, 0.05)   # d = 0.8 > 0.7
    assert len(src) == 0
    src, _ = m.get_correspondences([[0.1, 0.5, 0.5, 40]], 0.81, 0.05)
    assert len(src) == 1


def test_first_minimum_wins_in_enumeration_order(oracle):
    # two equidistant candidates in different voxels: x-outer enumeration visits voxel -1 first
    m = oracle.Map(1.0, 100.0)
    m.add_points([[1.25, 0.5, 0.5, 7], [-0.25, 0.5, 0.5, 7]])     # voxels (1,0,0) and (0,0,0)
    _, tgt = m.get_correspondences([[0.5, 0.5, 0.5, 7]], 6.0, 0.4)
    assert tgt[0, 0] == -0.25        # voxel 0 precedes voxel 1
    # same voxel: insertion order decides
    m = oracle.Map(1.0, 100.0)
    m.add_points([[0.75, 0.5, 0.5, 7], [0.25, 0.5, 0.5, 7]])
    _, tgt = m.get_correspondences([[0.5, 0.5, 0.5, 7]], 6.0, 0.4)
    assert tgt[0, 0] == 0.75


def test_empty_neighbourhood_is_rejected(oracle):
    m = oracle.Map(1.0, 100.0)
    m.add_points([[10.5, 0.5, 0.5, 7]])
    src, tgt = m.get_correspondences([[0.5, 0.5, 0.5, 7]], 100.0, 0.4)   # hazard H1
    assert len(src) == 0


def test_negative_coordinates_and_zero_straddle(oracle):
    mp, q = _random_scene(30, span=2.5)
    m = oracle.Map(1.0, 100.0)
    pm = pyref.PyMap(1.0, 100.0)
    m.add_points(mp)
    pm.add_points(mp)
    _, tgt, idx = m.get_correspondences(q, 6.0, 0.4, with_index=True)
    ref = pm.get_correspondences(q, 6.0, 0.4)
    assert [r[0] for r in ref] == list(idx)
    assert np.array_equal(tgt, np.array([r[1] for r in ref]))


# ---------------------------------------------------------------- KAT-6 RegisterFrame
def test_register_frame_empty_map_returns_guess(oracle):
    m = oracle.Map(1.0, 100.0)
    g = oracle.se3_exp(np.array([1, 2, 3, 0.1, 0.2, 0.3]))
    T, st = m.register_frame(np.zeros((5, 4)), g, 6.0, 0.6, 0.4)
    assert np.array_equal(T, g) and st.iterations == 0


def test_register_frame_recovers_planted_pose(oracle):
    rng = np.random.default_rng(40)
    mp = rng.uniform(-15, 15, size=(20000, 4))
    mp[:, 2] = rng.uniform(-2, 3, size=len(mp))
    mp[:, 3] = rng.choice([40, 50, 70, 80], size=len(mp))
    m = oracle.Map(1.0, 100.0)
    m.add_points(mp)
    kept = m.pointcloud()
    T_gt = oracle.se3_exp(np.array([0.2, -0.1, 0.05, 0.004, -0.003, 0.02]))
    scan_world = kept[rng.choice(len(kept), 3000, replace=False)]
    scan = oracle.transform_points(oracle.se3_inv(T_gt), scan_world)   # exact correspondences exist
    T, st = m.register_frame(scan, oracle.IDENTITY, 6.0, 2.0 / 3.0, 0.4)
    assert st.converged == 1 and st.iterations < 50
    err = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(T_gt), T))
    assert np.linalg.norm(err[:3]) < 1e-6 and np.linalg.norm(err[3:]) < 1e-6


def test_register_frame_no_correspondence_stops_after_one_iteration(oracle):
    m = oracle.Map(1.0, 100.0)
    m.add_points([[50.5, 0.5, 0.5, 1]])
    T, st = m.register_frame(np.array([[0.5, 0.5, 0.5, 1.0]]), oracle.IDENTITY, 1.0, 0.3, 0.4)
    assert st.iterations == 1 and st.converged == 1 and np.allclose(T, oracle.IDENTITY)


# ---------------------------------------------------------------- KAT-7 shard invariance
@pytest.mark.parametrize("shards", [2, 4, 8])
def test_shard_partials_sum_to_unsharded(oracle, shards):
    rng = np.random.default_rng(50)
    src = rng.normal(size=(4001, 4)) * 40
    tgt = src + rng.normal(size=(4001, 4)) * 0.2
    _, J, r = oracle.align_clouds(src, tgt, 0.4)
    Js, rs = np.zeros((6, 6)), np.zeros(6)
    per = -(-len(src) // shards)
    for k in range(shards):
        _, Jk, rk = oracle.align_clouds(src[k * per:(k + 1) * per], tgt[k * per:(k + 1) * per], 0.4)
        Js += Jk
        rs += rk
    assert np.allclose(Js, J, rtol=1e-12) and np.allclose(rs, r, rtol=1e-10, atol=1e-9)
