"""Trajectory metrics (row f-4): sageicp_metrics_* against analytic cases and the numpy
restatement of metrics/Metrics.cpp (oracle/metrics_ref.py).  CPU only."""
import numpy as np
import pytest


def _pose(yaw, t, pitch=0.0, roll=0.0):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T


def _trajectory(n=1200, step=1.0, yaw_rate=0.002, seed=0):
    rng = np.random.default_rng(seed)
    T = np.eye(4)
    out = [T.copy()]
    for _ in range(n - 1):
        T = T @ _pose(yaw_rate + 1e-4 * rng.standard_normal(), [step, 0.01 * rng.standard_normal(), 0.0],
                      1e-4 * rng.standard_normal())
        out.append(T.copy())
    return np.array(out)


@pytest.fixture(scope="module")
def ref():
    from oracle import metrics_ref
    return metrics_ref


def test_identity_has_zero_error(sage):
    gt = _trajectory()
    t, r = sage.seq_error(gt, gt)
    assert abs(t) < 1e-9 and abs(r) < 1e-4     # acos near 1 amplifies rounding to ~1e-8 rad
    a, b = sage.absolute_trajectory_error(gt, gt)
    assert a < 1e-6 and b < 1e-9


def _segment_ratio(n):
    """mean over the devkit's segments of (frames spanned) / (nominal length) on a 1 m/frame line:
    a segment ends at the first frame BEYOND its length, i.e. spans L + 1 metres"""
    r = [(L + 1) / L for first in range(0, n, 10) for L in (100, 200, 300, 400, 500, 600, 700, 800)
         if first + L + 1 <= n - 1]
    return float(np.mean(r))


def test_seq_error_pure_scale_drift(sage):
    """an estimate that moves 1 % too far per step has 1 % translational error per metre travelled"""
    gt = np.array([_pose(0.0, [float(i), 0.0, 0.0]) for i in range(1500)])
    res = np.array([_pose(0.0, [1.01 * i, 0.0, 0.0]) for i in range(1500)])
    t, r = sage.seq_error(gt, res)
    assert t == pytest.approx(1.0 * _segment_ratio(1500), rel=1e-5) and abs(r) < 1e-4


def test_seq_error_pure_heading_drift(sage):
    """a constant yaw-rate error of w rad/m gives w rad/m: 100 w / 3.14 * 180 in the reference's units"""
    w = 1e-4
    gt = np.array([_pose(0.0, [float(i), 0.0, 0.0]) for i in range(1500)])
    res, T = [], np.eye(4)
    for _ in range(1500):
        res.append(T.copy())
        T = T @ _pose(w, [1.0, 0.0, 0.0])
    t, r = sage.seq_error(gt, np.array(res))
    expect = w * _segment_ratio(1500) * 100 / 3.14 * 180
    assert r == pytest.approx(expect, rel=1e-4)


def test_too_short_sequence_is_nan_like_the_reference(sage):
    gt = _trajectory(n=50)
    t, r = sage.seq_error(gt, gt)
    assert np.isnan(t) and np.isnan(r)


def test_ate_is_invariant_to_a_rigid_motion_of_the_estimate(sage):
    """a known similarity (rigid): the alignment removes it, the translation error vanishes, the
    rotation error is the angle of the applied rotation"""
    gt = _trajectory(seed=3)
    M = _pose(0.3, [5.0, -2.0, 1.0], pitch=0.1, roll=-0.2)
    res = np.array([M @ T for T in gt])
    a, b = sage.absolute_trajectory_error(gt, res)
    assert b < 1e-6
    assert a < 1e-6          # the aligned estimate coincides with the ground truth, rotation included


def test_ate_known_offsets(sage):
    """positions perturbed by a zero-mean, alignment-neutral pattern: RMSE known in closed form"""
    n = 400
    gt = np.array([_pose(0.0, [float(i), 0.0, 0.0]) for i in range(n)])
    res = gt.copy()
    res[:, 2, 3] += 0.05 * np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * np.where(np.arange(n) // 2 % 2 == 0, 1.0, -1.0)
    a, b = sage.absolute_trajectory_error(gt, res)
    assert b == pytest.approx(0.05, rel=1e-3) and a < 1e-4


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_against_numpy_restatement(sage, ref, seed):
    rng = np.random.default_rng(seed)
    gt = _trajectory(n=1400, yaw_rate=0.003, seed=seed)
    res = []
    D = np.eye(4)
    for T in gt:
        D = D @ _pose(2e-5 * rng.standard_normal(), 2e-3 * rng.standard_normal(3), 1e-5 * rng.standard_normal())
        res.append(T @ D)
    res = np.array(res)
    t, r = sage.seq_error(gt, res)
    et, er = ref.seq_error(gt, res)
    assert t == pytest.approx(float(et), rel=1e-5) and r == pytest.approx(float(er), rel=1e-5)
    a, b = sage.absolute_trajectory_error(gt, res)
    ea, eb = ref.absolute_trajectory_error(gt, res)
    assert a == pytest.approx(float(ea), rel=1e-4, abs=1e-7) and b == pytest.approx(float(eb), rel=1e-5)


def test_ate_planar_trajectory_rank_deficient_covariance(sage, ref):
    """all positions in a plane: the 3x3 covariance has a zero singular value (Umeyama's reflection
    guard decides the third axis)"""
    gt = _trajectory(n=300, seed=5)
    gt[:, 2, 3] = 0.0
    M = _pose(0.7, [1.0, 2.0, 0.0])
    res = np.array([M @ T for T in gt])
    a, b = sage.absolute_trajectory_error(gt, res)
    ea, eb = ref.absolute_trajectory_error(gt, res)
    assert b == pytest.approx(float(eb), abs=1e-6) and b < 1e-6
