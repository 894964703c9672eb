"""Property tests (hypothesis) of the oracle against the independent pure-Python restatement:
random clouds, voxel sizes, thresholds and map capacities for the semantic NN (KAT-4) and shard
additivity of the Gauss-Newton sums (KAT-7)."""
import os

import numpy as np
from hypothesis import given, settings, strategies as st

import pyref

LABELS = [0, 0, 10, 40, 44, 50, 70, 71, 80]


def _examples(n):
    """SAGE_TEST_EXAMPLES=k runs k times the usual number of random examples (campaigns; profiles/README.md)"""
    return n * max(1, int(os.environ.get("SAGE_TEST_EXAMPLES", "1")))


@settings(max_examples=_examples(25), deadline=None)
@given(seed=st.integers(0, 2**31 - 1), vs=st.sampled_from([0.3, 0.8, 1.0, 2.5]),
       th=st.sampled_from([0.05, 0.4, 1.0, 1.7]), md=st.sampled_from([0.2, 0.9, 6.0]),
       basic=st.integers(0, 5), critical=st.integers(1, 4), span=st.sampled_from([1.5, 4.0, 9.0]))
def test_get_correspondences_matches_python(oracle, seed, vs, th, md, basic, critical, span):
    rng = np.random.default_rng(seed)
    mp = rng.uniform(-span, span, size=(400, 4))
    mp[:, 3] = rng.choice(LABELS, size=len(mp))
    q = rng.uniform(-span - vs, span + vs, size=(60, 4))
    q[:, 3] = rng.choice(LABELS, size=len(q))
    if seed % 3 == 0:                       # queries exactly on voxel faces
        q[:, :3] = np.round(q[:, :3] / vs) * vs
    m = oracle.Map(vs, 100.0, basic=basic, critical=critical)
    pm = pyref.PyMap(vs, 100.0, basic=basic, critical=critical)
    m.add_points(mp)
    pm.add_points(mp)
    assert m.size() == pm.size()
    _, tgt, idx = m.get_correspondences(q, md, th, nthreads=2, with_index=True)
    ref = pm.get_correspondences(q, md, th)
    assert [r[0] for r in ref] == list(idx)
    if ref:
        assert np.array_equal(tgt, np.array([r[1] for r in ref]))
    assert m.last_sum_candidates == pm.last_candidates


@settings(max_examples=_examples(20), deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 300), shards=st.sampled_from([2, 3, 4, 8]),
       kernel=st.sampled_from([0.1, 0.6667, 3.0]))
def test_gauss_newton_sums_are_shard_additive(oracle, seed, n, shards, kernel):
    rng = np.random.default_rng(seed)
    src = rng.normal(size=(n, 4)) * 50
    tgt = src + rng.normal(size=(n, 4)) * 0.3
    _, J, r = oracle.align_clouds(src, tgt, kernel)
    eJ, er = pyref.normal_equations(src, tgt, kernel)
    scale = max(1.0, np.abs(eJ).max())
    assert np.allclose(J, eJ, rtol=1e-11, atol=1e-12 * scale)
    assert np.allclose(r, er, rtol=1e-9, atol=1e-12 * scale)
    Js, rs = np.zeros((6, 6)), np.zeros(6)
    per = -(-n // shards)
    for k in range(shards):
        _, Jk, rk = oracle.align_clouds(src[k * per:(k + 1) * per], tgt[k * per:(k + 1) * per], kernel)
        Js += Jk
        rs += rk
    assert np.allclose(Js, J, rtol=1e-11, atol=1e-12 * scale)
    assert np.allclose(rs, r, rtol=1e-9, atol=1e-12 * scale)
