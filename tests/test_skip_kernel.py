"""k_skip (kernels.hip): queries that provably keep their previous answer are not searched.  The rule is
exact — a kept answer is the answer a search would give — so a registration through k_skip must follow
the oracle's iteration for iteration: same correspondence counts, same number of iterations, the pose
within rounding (the Gauss-Newton sums are grouped differently: 1e-12, not bit-identical).

Needs a real MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

from test_loop_kernel import Env, _workload, pose_error

pytestmark = pytest.mark.gpu


def _run(gpu_sage, w, p, init=None, **env):
    init = gpu_sage.IDENTITY if init is None else init
    with Env(SAGEICP_LOOP=0, **env):
        return gpu_sage.register_frame(w["scan"], w["map"], init, p["max_dist"], p["kernel"], p["sem_th"],
                                       return_stats=True)


@pytest.mark.parametrize("name,scale,params", [
    ("c2", 0.1, "cold"), ("c2", 0.1, "steady"), ("c2", 0.5, "cold"), ("c4", 0.1, "steady"),
    ("c5", 0.1, "dense"), ("c5", 0.1, "dense_nosem"), ("c1", 1.0, "cold"),
])
def test_skip_search_follows_the_oracle(gpu_sage, oracle, name, scale, params, scan_form):
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, name, scale)
    p = syn.PARAMS[params]
    plain, sp = _run(gpu_sage, w, p, SAGEICP_SKIP=0)
    pose, st = _run(gpu_sage, w, p, SAGEICP_SKIP=1)
    assert sp.skip_search == 0 and st.skip_search == 1
    n = len(w["scan"])
    assert sp.queries_searched == sp.iterations * n
    assert 0 < st.queries_searched < 0.8 * st.iterations * n, "nothing was kept"
    assert st.iterations == sp.iterations and st.converged == sp.converged == 1
    assert list(st.n_corr_hist) == list(sp.n_corr_hist)
    assert st.sum_candidates == sp.sum_candidates             # C_q of kept queries is accounted from their state
    assert st.pairs_evaluated < sp.pairs_evaluated
    assert np.allclose(pose, plain, rtol=0.0, atol=1e-10)
    opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, pose)
    assert dt < 1e-7 and dr < 1e-7
    assert st.iterations == ost.iterations and st.n_corr_first == ost.n_corr_first and st.n_corr_last == ost.n_corr_last
    assert st.sum_candidates == ost.sum_candidates_total


@pytest.mark.parametrize("lw", [0, 1, 2, 3, 4])
def test_skip_search_in_every_lanes_per_query_variant(gpu_sage, oracle, lw, scan_form):
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    p = syn.PARAMS["cold"]
    pose, st = _run(gpu_sage, w, p, SAGEICP_SKIP=1, SAGEICP_LW=lw)
    assert st.skip_search == 1 and st.lanes_per_query == 1 << lw
    opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, pose)
    assert dt < 1e-7 and dr < 1e-7 and st.iterations == ost.iterations
    assert st.n_corr_first == ost.n_corr_first and st.n_corr_last == ost.n_corr_last
    assert st.sum_candidates == ost.sum_candidates_total


@pytest.mark.parametrize("mm", [1, 10, 200, 2000])
def test_margin_cap_changes_the_work_not_the_result(gpu_sage, oracle, mm):
    """the cap on the keep-margin (how far beyond its answer a search looks) trades searches against
    work per search; the registration is the same for every value"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.1)
    p = syn.PARAMS["cold"]
    ref, sr = _run(gpu_sage, w, p, SAGEICP_SKIP=0)
    pose, st = _run(gpu_sage, w, p, SAGEICP_SKIP=1, SAGEICP_SKIP_MARGIN_MM=mm)
    assert st.skip_search == 1 and st.iterations == sr.iterations
    assert list(st.n_corr_hist) == list(sr.n_corr_hist)
    assert np.allclose(pose, ref, rtol=0.0, atol=1e-10)


def test_skip_search_with_near_ties_voxel_crossings_and_a_guess(gpu_sage, oracle):
    """a scene built to stress the rule: clusters of map points 1e-9 m apart (margins of ~0: never kept),
    exact duplicates (ties: margin 0), queries that cross voxel faces while the pose creeps"""
    rng = np.random.default_rng(77)
    base = rng.uniform(-6, 6, size=(4000, 4))
    base[:, 2] = rng.uniform(-1, 1, 4000)
    base[:, 3] = rng.choice([0, 40, 50, 70], 4000)
    near = base[:1500].copy()
    near[:, :3] += rng.uniform(-1e-9, 1e-9, size=(1500, 3))
    dup = base[1500:2500].copy()
    mp = np.concatenate([base, near, dup, rng.uniform(-6, 6, size=(20000, 4)) * [1, 1, 0.2, 0] + [0, 0, 0, 40]])
    from sage_icp_amd import synthetic as syn
    T = syn.pose_from_rpy_t([0.3, -0.2, 1.2], [0.21, -0.17, 0.05])
    q = syn.apply_pose(syn.invert_pose(T), mp[rng.choice(len(mp), 6000, replace=False)] + [0.003, -0.002, 0.001, 0])
    a = gpu_sage.VoxelHashMap(1.0, 100.0)
    b = oracle.Map(1.0, 100.0)
    a.AddPoints(mp)
    b.add_points(mp)
    w = dict(map=a, scan=np.ascontiguousarray(q))
    for th, md, k in ((0.4, 3.0, 0.5), (1.0, 1.0, 0.1), (2.5, 2.0, 0.3)):
        p = dict(max_dist=md, kernel=k, sem_th=th)
        pose, st = _run(gpu_sage, w, p, SAGEICP_SKIP=1)
        opose, ost = b.register_frame(w["scan"], oracle.IDENTITY, md, k, th)
        dt, dr = pose_error(oracle, opose, pose)
        assert st.skip_search == 1 and dt < 1e-7 and dr < 1e-7 and st.iterations == ost.iterations
        assert st.n_corr_first == ost.n_corr_first and st.n_corr_last == ost.n_corr_last


def test_unusable_sem_th_turns_the_rule_off(gpu_sage, oracle):
    """a negative sem_th gives the search no lower bound (a larger distance can scale to a smaller one):
    no pruning, no keeping — k_icp runs"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    pose, st = _run(gpu_sage, w, dict(max_dist=2.0, kernel=0.3, sem_th=-0.5), SAGEICP_SKIP=1)
    assert st.skip_search == 0
