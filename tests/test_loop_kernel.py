"""The one-launch ICP loop (k_loop, kernels.hip) against the launch-per-iteration loop (k_icp + k_fin)
and the oracle.  Both loops evaluate the same arithmetic on the same data; the pair terms of every block
of four consecutive queries are added in a fixed order and become exact fixed-point numbers, everything
after that is integer addition: the poses must be BIT-identical whatever the lanes per query, the shape
of the launch or the loop; the oracle comparison is the usual parity bar.

Needs a real MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def pose_error(oracle, A, B):
    e = oracle.se3_log(oracle.se3_mul(oracle.se3_inv(A), B))
    return np.linalg.norm(e[:3]), np.linalg.norm(e[3:])


class Env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _workload(gpu_sage, oracle, name, scale):
    from sage_icp_amd import synthetic as syn
    w = syn.make_workload(name, lambda: gpu_sage.VoxelHashMap(syn.WORKLOADS[name]["voxel"], 100.0), scale=scale)
    om = oracle.Map(w["voxel"], 100.0)
    om.add_points(w["stream"])
    return w, om


def _both_loops(gpu_sage, w, p, init=None, **env):
    """the one-launch loop as the library shapes it (or as `env` forces it), then the launch-per-iteration loop
    as the library shapes IT — other lanes per query as a rule: the Gauss-Newton sums are exact integers from
    the blocks of four consecutive queries on, so the bits must not depend on any of that"""
    init = gpu_sage.IDENTITY if init is None else init
    with Env(SAGEICP_LOOP=2, **env):
        b, sb = gpu_sage.register_frame(w["scan"], w["map"], init, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    env = {k: v for k, v in env.items() if k not in ("SAGEICP_LW",)}
    with Env(SAGEICP_LOOP=0, **env):
        a, sa = gpu_sage.register_frame(w["scan"], w["map"], init, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    assert sa.single_launch == 0
    return a, sa, b, sb


def _same(sa, sb):
    assert sa.iterations == sb.iterations and sa.converged == sb.converged
    assert sa.n_corr_first == sb.n_corr_first and sa.n_corr_last == sb.n_corr_last
    assert list(sa.n_corr_hist) == list(sb.n_corr_hist)
    assert sa.last_step_norm == sb.last_step_norm
    assert sa.sum_candidates == sb.sum_candidates


@pytest.mark.parametrize("name,scale,params", [
    ("c1", 1.0, "cold"), ("c2", 0.1, "cold"), ("c2", 0.1, "steady"), ("c2", 0.25, "cold"),
    ("c5", 0.05, "dense"), ("c5", 0.05, "dense_nosem"), ("c4", 0.05, "steady"),
])
def test_one_launch_loop_is_bit_identical_and_matches_oracle(gpu_sage, oracle, name, scale, params):
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, name, scale)
    p = syn.PARAMS[params]
    a, sa, b, sb = _both_loops(gpu_sage, w, p)
    assert sb.single_launch == 1, "the frame was expected to fit the one-launch loop"
    assert np.array_equal(a, b)
    _same(sa, sb)
    opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, b)
    assert dt < 1e-7 and dr < 1e-7
    assert sb.iterations == ost.iterations and sb.converged == ost.converged
    assert sb.n_corr_first == ost.n_corr_first and sb.n_corr_last == ost.n_corr_last
    assert sb.sum_candidates == ost.sum_candidates_total


@pytest.mark.parametrize("lw", [1, 2, 3, 4])
@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("waves", [1, 3, 8])
def test_every_shape_of_the_one_launch_loop(gpu_sage, oracle, lw, compact, waves):
    """lanes per query x scan form x waves per workgroup (incl. counts that are not powers of two)"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    p = syn.PARAMS["cold"]
    a, sa, b, sb = _both_loops(gpu_sage, w, p, SAGEICP_LW=lw, SAGEICP_FILTER=compact, SAGEICP_LOOP_WAVES=waves)
    assert sb.single_launch == 1 and sb.lanes_per_query == 1 << lw and sb.compact_scan == compact
    assert np.array_equal(a, b)
    _same(sa, sb)
    opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, b)
    assert dt < 1e-7 and dr < 1e-7 and sb.iterations == ost.iterations


@pytest.mark.parametrize("name,scale,params", [("c2", 0.1, "cold"), ("c5", 0.05, "dense"), ("c1", 1.0, "cold")])
def test_bits_do_not_depend_on_lanes_per_query_or_loop(gpu_sage, oracle, name, scale, params):
    """1, 2, 4, 8, 16 lanes per query through the launch-per-iteration loop, 2..16 through the one-launch loop,
    with and without the compact scan: ONE pose, to the bit (round 4: equal only at equal lanes per query)"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, name, scale)
    p = syn.PARAMS[params]
    poses = []
    for loop in (0, 2):
        for lw in range(0 if loop == 0 else 1, 5):
            for compact in (0, 1):
                with Env(SAGEICP_LOOP=loop, SAGEICP_LW=lw, SAGEICP_FILTER=compact):
                    a, sa = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"],
                                                    p["sem_th"], return_stats=True)
                assert sa.lanes_per_query == 1 << lw and sa.single_launch == (1 if loop else 0)
                poses.append((a, sa))
    for a, sa in poses[1:]:
        assert np.array_equal(a, poses[0][0])
        _same(sa, poses[0][1])


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 4097])
def test_ragged_frame_sizes(gpu_sage, oracle, n):
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    w = dict(w, scan=np.ascontiguousarray(w["scan"][:n]))
    p = syn.PARAMS["cold"]
    a, sa, b, sb = _both_loops(gpu_sage, w, p)
    assert sb.single_launch == 1
    assert np.array_equal(a, b)
    _same(sa, sb)
    if n < 64:
        # one or two pairs leave the 6x6 system rank deficient (pivots of ~1e-17 instead of exact
        # zeros): the step is rounding noise divided by rounding noise, on the CPU as on the GPU —
        # there is no parity to speak of, only the agreement of the two loops above
        return
    opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, b)
    assert dt < 1e-6 and dr < 1e-6 and sb.iterations == ost.iterations


def test_profiling_of_the_one_launch_loop(gpu_sage, oracle):
    """with profiling on, the launch is bracketed by HIP events: us_nn is the whole loop, nn_launches the
    iterations (so that us_nn / nn_launches stays a time per iteration), us_fin stays zero"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    p = syn.PARAMS["cold"]
    for level in (1, 2):
        gpu_sage.set_profiling(level)
        try:
            with Env(SAGEICP_LOOP=2):
                _, st = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"],
                                                p["sem_th"], return_stats=True)
        finally:
            gpu_sage.set_profiling(0)
        assert st.single_launch == 1 and st.nn_launches == st.iterations and st.us_fin == 0
        assert 2.0 < st.us_nn / st.nn_launches < 200.0
        assert 0 < st.pairs_evaluated <= st.sum_candidates


def test_no_correspondence_and_far_frames(gpu_sage, oracle):
    """a frame with no map point in reach: zero pairs, the zero system solves to a zero step, the loop
    ends after one iteration with the guess (Registration.cpp:92,137)"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    far = w["scan"].copy()
    far[:, :3] += 5000.0
    w = dict(w, scan=far)
    p = syn.PARAMS["cold"]
    a, sa, b, sb = _both_loops(gpu_sage, w, p)
    assert sb.single_launch == 1 and sb.iterations == 1 and sb.n_corr_first == 0
    assert np.array_equal(a, b) and np.array_equal(b, gpu_sage.IDENTITY)


def test_initial_guess_and_repeated_calls(gpu_sage, oracle):
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.1)
    p = syn.PARAMS["steady"]
    guess = syn.pose_from_rpy_t([0.05, -0.02, 0.8], [0.45, 0.12, 0.0])
    a, sa, b, sb = _both_loops(gpu_sage, w, p, init=guess)
    assert sb.single_launch == 1 and np.array_equal(a, b)
    _same(sa, sb)
    opose, ost = om.register_frame(w["scan"], guess, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, b)
    assert dt < 1e-7 and dr < 1e-7 and sb.iterations == ost.iterations
    # the shared block is re-initialised before every launch: the same call again, and again
    with Env(SAGEICP_LOOP=2):
        for _ in range(3):
            c = gpu_sage.register_frame(w["scan"], w["map"], guess, p["max_dist"], p["kernel"], p["sem_th"])
            assert np.array_equal(b, c)


def test_frame_that_does_not_fit_uses_the_launch_per_iteration_loop(gpu_sage, oracle):
    """rows and per-query state of every query have to fit the LDS of the machine (232 B per query, 160 KB per
    CU: ~170k queries on 256 CUs)"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.1)
    big = np.ascontiguousarray(np.tile(w["scan"], (20, 1)))         # 240k points
    p = syn.PARAMS["steady"]
    with Env(SAGEICP_LOOP=2, SAGEICP_LW=3):
        _, st = gpu_sage.register_frame(big, w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    assert st.single_launch == 0 and st.converged == 1


@pytest.mark.parametrize("lw,waves,gpw", [(2, 7, 3), (3, 4, 9), (2, 8, 2), (4, 5, 7), (1, 7, 2), (2, 4, 5), (2, 4, 8), (3, 2, 3), (4, 4, 6)])
def test_several_groups_per_wave(gpu_sage, oracle, lw, waves, gpw):
    """a workgroup that owns more groups of queries than it has waves: its waves take them one after another from
    an LDS counter (the shape the headline frame runs in) — the same bits as one wave per group"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.1)
    p = syn.PARAMS["cold"]
    a, sa, b, sb = _both_loops(gpu_sage, w, p, SAGEICP_LW=lw, SAGEICP_LOOP_WAVES=waves, SAGEICP_LOOP_GPW=gpw)
    assert sb.single_launch == 1 and sb.lanes_per_query == 1 << lw
    assert np.array_equal(a, b)
    _same(sa, sb)
    with Env(SAGEICP_LOOP=2, SAGEICP_LW=lw, SAGEICP_LOOP_WAVES=waves, SAGEICP_LOOP_GPW=gpw, SAGEICP_LOOP_CONTIGUOUS=1):
        c, sc = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    assert sc.single_launch == 1 and np.array_equal(a, c)       # (one contiguous range of the frame per XCD)
    opose, ost = om.register_frame(w["scan"], oracle.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"])
    dt, dr = pose_error(oracle, opose, b)
    assert dt < 1e-7 and dr < 1e-7 and sb.iterations == ost.iterations


def test_headline_frame_in_one_launch(gpu_sage, oracle):
    """c2 at full size (120k queries against the 1M-point map) fits the one-launch loop as the library shapes it,
    bit-identical to the launch-per-iteration loop at the same lanes per query"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 1.0)
    p = syn.PARAMS["cold"]
    with Env(SAGEICP_LOOP=1):
        b, sb = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    assert sb.single_launch == 1, "the headline frame was expected to run in one launch"
    with Env(SAGEICP_LOOP=0):
        a, sa = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    assert np.array_equal(a, b)
    _same(sa, sb)
    assert sb.iterations == 185 and sb.converged == 1          # (the golden count of this scene, BENCH r01-r04)


def test_timeout_inside_the_launch_falls_back(gpu_sage, oracle):
    """SAGEICP_LOOP_TIMEOUT_TICKS=1: every wait inside the launch gives up at once; the kernel must end
    (no hang) and the frame must still be registered — by the other loop, with the lanes per query the one-launch
    loop would have taken: the same pose to the bit whether the launch succeeds, times out, or is not tried"""
    from sage_icp_amd import synthetic as syn
    w, om = _workload(gpu_sage, oracle, "c2", 0.05)
    p = syn.PARAMS["cold"]
    with Env(SAGEICP_LOOP=2):
        a, sa = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"], p["sem_th"],
                                        return_stats=True)
    assert sa.single_launch == 1
    s0 = w["map"].loop_status()
    assert (s0.last_fallback, s0.timeouts, s0.cooldown_calls) == (0, 0, 0) and s0.calls_single_launch >= 1
    with Env(SAGEICP_LOOP=2, SAGEICP_LOOP_TIMEOUT_TICKS=1, SAGEICP_LOOP_COOLDOWN=2):
        b, sb = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"],
                                        p["sem_th"], return_stats=True)
    assert np.array_equal(a, b)
    # (a tick is 10 ns: a wait that long never succeeds on a grid of this size)
    assert sb.single_launch == 0
    # the handle says what happened (sageicp_map_loop_status): one launch gave up, two calls of cool-down ahead
    s1 = w["map"].loop_status()
    assert (s1.last_fallback, s1.timeouts, s1.cooldown_calls) == (1, 1, 2)         # SAGEICP_LOOP_FALLBACK_TIMEOUT
    assert s1.calls_per_iteration == s0.calls_per_iteration + 1 and s1.calls_single_launch == s0.calls_single_launch
    # a map whose launch timed out stays away from k_loop for a while (here: two calls), then tries again
    with Env(SAGEICP_LOOP=2):
        forms = []
        for _ in range(3):
            c, sc = gpu_sage.register_frame(w["scan"], w["map"], gpu_sage.IDENTITY, p["max_dist"], p["kernel"],
                                            p["sem_th"], return_stats=True)
            assert np.array_equal(a, c)
            forms.append((sc.single_launch, w["map"].loop_status().last_fallback))
    assert forms == [(0, 2), (0, 2), (1, 0)]                                        # ... _COOLDOWN twice, then _NONE


def test_streamed_frames_through_the_pipeline(gpu_sage, oracle):
    """the per-frame pipeline (sageicp_pipeline_*) registers 24k-point sources: both loops, same poses"""
    from sage_icp_amd import synthetic as syn
    frames, _ = syn.make_stream(7, 8, points_per_frame=60000)
    poses = {}
    for mode in (0, 2):
        with Env(SAGEICP_LOOP=mode, SAGEICP_LW=3):        # (the same lanes per query in both: see _both_loops)
            pipe = gpu_sage.SageICP()
            out = [pipe.RegisterFrame(f) for f in frames]
            poses[mode] = [o[0].copy() for o in out]
            if mode == 2:
                assert all(o[4].single_launch == 1 for o in out[1:]), "streamed sources fit the one-launch loop"
    for a, b in zip(poses[0], poses[2]):
        assert np.array_equal(a, b)
