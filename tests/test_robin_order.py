"""The oracle's emulation of tsl::robin_map v1.0.1 iteration order (oracle/sage_oracle.cpp,
RobinOrder) against a second, independent restatement in Python, and the three places the
reference's results depend on that order.  CPU only."""
import numpy as np
import pytest


def vhash(k):
    """core/VoxelHashMap.hpp:72-77"""
    a, b, c = (int(v) & 0xFFFFFFFF for v in k)
    return ((a * 73856093) ^ (b * 19349663) ^ (c * 83492791)) & 0xFFFFFFFF & ((1 << 20) - 1)


class PyRobin:
    """robin-hood linear probing, power-of-two growth from 0 at load 0.5, backward-shift erase —
    written from the published description of tsl::robin_map, independently of the C++ one"""

    def __init__(self):
        self.slots = []          # None or [key, value, dist]
        self.n = 0
        self.max_dist = 0        # largest distance any entry was ever stored with

    def _insert_raw(self, key, val):
        mask = len(self.slots) - 1
        i, cur = vhash(key) & mask, [key, val, 0]
        while True:
            r = self.slots[i]
            if r is None:
                self.slots[i] = cur
                self.max_dist = max(self.max_dist, cur[2])
                return
            if cur[2] > r[2]:
                self.slots[i], cur = cur, r
                self.max_dist = max(self.max_dist, self.slots[i][2])
            cur[2] += 1
            i = (i + 1) & mask

    def insert(self, key, val):
        if self.n >= int(np.float32(len(self.slots)) * np.float32(0.5)):
            old = [s for s in self.slots if s is not None]      # bucket order
            self.slots = [None] * (2 if not self.slots else 2 * len(self.slots))
            for k, v, _ in old:
                self._insert_raw(k, v)
        self._insert_raw(key, val)
        self.n += 1

    def find(self, key):
        if not self.slots:
            return -1
        mask = len(self.slots) - 1
        i, d = vhash(key) & mask, 0
        while self.slots[i] is not None and d <= self.slots[i][2]:
            if self.slots[i][0] == key:
                return i
            i, d = (i + 1) & mask, d + 1
        return -1

    def erase(self, key):
        i = self.find(key)
        if i < 0:
            return
        mask = len(self.slots) - 1
        self.slots[i] = None
        self.n -= 1
        j = (i + 1) & mask
        while self.slots[j] is not None and self.slots[j][2] > 0:
            self.slots[j][2] -= 1
            self.slots[i], self.slots[j] = self.slots[j], None
            i, j = j, (j + 1) & mask

    def order(self):
        return [s[1] for s in self.slots if s is not None]


def test_growth_schedule(oracle):
    """0 -> 2 -> 4 -> 8 ...: an insertion grows the array when size >= buckets / 2"""
    keys = np.array([[i, 0, 0] for i in range(40)])
    for n, buckets in [(1, 2), (2, 4), (3, 8), (4, 8), (5, 16), (8, 16), (9, 32), (17, 64), (33, 128)]:
        order, bc = oracle.robin_order_of(keys[:n])
        assert bc == buckets and sorted(order) == list(range(n))


@pytest.mark.parametrize("seed", range(6))
def test_order_matches_independent_restatement(oracle, seed):
    rng = np.random.default_rng(seed)
    span = [3, 30, 300, 3000, 40, 8][seed]
    keys = np.unique(rng.integers(-span, span, size=(1500, 3)), axis=0)
    rng.shuffle(keys)
    erase = keys[rng.choice(len(keys), len(keys) // 3, replace=False)]
    r = PyRobin()
    for i, k in enumerate(keys):
        r.insert(tuple(k), i)
    got, bc = oracle.robin_order_of(keys)
    assert list(got) == r.order() and bc == len(r.slots)
    for k in erase:
        r.erase(tuple(k))
    got, _ = oracle.robin_order_of(keys, erase)
    assert list(got) == r.order()


def test_voxel_downsample_order_switch(oracle):
    """same survivors either way; arrival order by default, bucket order with the switch"""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-20, 20, size=(5000, 4))
    pts[:, 3] = rng.choice([0, 40, 50, 70, 71], size=5000)
    labels, sizes = [[40, 44], [50], [70, 71], [0]], [0.6, 1.0, 0.9, 1.0]
    oracle.set_robin_order(False)
    try:
        a = oracle.voxel_downsample(pts, labels, sizes, 1.0)
        oracle.set_robin_order(1)
        b = oracle.voxel_downsample(pts, labels, sizes, 1.0)
    finally:
        oracle.set_robin_order(False)
    assert a.shape == b.shape and not np.array_equal(a, b)
    assert sorted(map(tuple, a)) == sorted(map(tuple, b))
    # group by group, b is the PyRobin order of a's voxels
    off = 0
    for g, vs in zip(labels, sizes):
        sel = a[np.isin(a[:, 3].astype(int), g)]
        r = PyRobin()
        for i, p in enumerate(sel):
            r.insert(tuple(int(v) for v in (p[:3] / vs)), i)
        assert np.array_equal(b[off:off + len(sel)], sel[r.order()])
        off += len(sel)


def test_far_voxel_sweep_skips_what_shifts_into_the_erased_bucket(oracle):
    """erase-while-iterating (VoxelHashMap.cpp:177-183): with the switch on, a far voxel that the
    backward shift moves into the bucket just erased survives this sweep and goes in the next"""
    rng = np.random.default_rng(5)
    pts = np.zeros((6000, 4))
    pts[:, :3] = rng.uniform(-60, 60, size=(6000, 3))
    pts[:, 3] = 40
    results = {}
    for on in (False, True):
        oracle.set_robin_order(3 if on else 0)
        try:
            m = oracle.Map(1.0, 30.0)
            m.add_points(pts)
            before = m.num_voxels()
            m.remove_far(np.zeros(3))
            first = m.num_voxels()
            m.remove_far(np.zeros(3))
            results[on] = (before, first, m.num_voxels(), m.pointcloud())
        finally:
            oracle.set_robin_order(False)
    b0, f0, s0, pc0 = results[False]
    b1, f1, s1, pc1 = results[True]
    assert b0 == b1 and f0 == s0                       # collect-then-erase finishes in one sweep
    assert f1 > f0                                     # the reference's sweep leaves some behind ...
    assert np.all(np.linalg.norm(pc0[:, :3], axis=1) <= 30.0 + 2.0)
    assert s1 < f1                                     # ... and takes (most of) them the next time
    assert sorted(map(tuple, pc0)) == sorted(map(tuple, pc1)) or s1 >= s0


def test_product_replay_matches_independent_restatement(sage):
    """the product's host replay (csrc/robin_order.hpp, through the C ABI; no device involved):
    same iteration order as the Python restatement, on key sets from 0 to 40k voxels, called
    back to back so that the reused bucket arrays are exercised with shrinking and growing sizes"""
    L = sage.lib()
    rng = np.random.default_rng(11)
    for n, span in [(0, 3), (1, 3), (2, 3), (3, 3), (700, 6), (5000, 40), (40000, 300), (33, 2), (9000, 12), (2049, 4000)]:
        keys = np.unique(rng.integers(-span, span, size=(max(n, 1) * 2, 3)), axis=0)
        rng.shuffle(keys)
        keys = np.ascontiguousarray(keys[:n].astype(np.int32))
        n = len(keys)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        assert L.sageicp_robin_iteration_order(keys.ctypes.data, n, out.ctypes.data) == 0
        r = PyRobin()
        for i, k in enumerate(keys):
            r.insert(tuple(int(v) for v in k), i)
        assert list(out[:n]) == r.order(), (n, span)


def _voxels_with_hash(pred, want, rng, span=400):
    """distinct voxels whose reference hash satisfies `pred` (vectorised search over a random cube)"""
    got = []
    seen = set()
    while len(got) < want:
        k = rng.integers(-span, span, size=(200000, 3)).astype(np.int64)
        h = ((k[:, 0] * 73856093) ^ (k[:, 1] * 19349663) ^ (k[:, 2] * 83492791)) & 0xFFFFF
        for row in k[pred(h)]:
            t = tuple(int(v) for v in row)
            if t not in seen:
                seen.add(t)
                got.append(t)
                if len(got) == want:
                    break
    return got


def test_product_replay_on_wrap_heavy_key_sets(sage):
    """key sets built to wrap around the end of the bucket array at every growth (hashes whose low
    bits are all ones, at the ends of both halves of the next array) — where the replay's fast
    growth path hands over to one-by-one insertion: still the restatement's order"""
    L = sage.lib()
    rng = np.random.default_rng(23)
    cases = [
        lambda h: (h & 0x3F) >= 0x3C,                       # the top 4 of every 64: wraps in the small arrays
        lambda h: (h & 0xFF) >= 0xF8,
        lambda h: ((h & 0x3FF) >= 0x3FA) | ((h & 0x3FF) == 0x1FF) | ((h & 0x3FF) < 2),
        lambda h: (h & 0xFFF) >= 0xFF0,
    ]
    for ci, pred in enumerate(cases):
        for n in (40, 300, 1500, 5000):
            hot = _voxels_with_hash(pred, n // 3, rng)
            cold = [tuple(int(v) for v in r) for r in rng.integers(-300, 300, size=(n, 3))]
            keys = list(dict.fromkeys(hot + cold))
            order = rng.permutation(len(keys))
            keys = np.ascontiguousarray(np.array([keys[i] for i in order], dtype=np.int32))
            m = len(keys)
            out = np.zeros(m, dtype=np.uint32)
            rc = L.sageicp_robin_iteration_order(keys.ctypes.data, m, out.ctypes.data)
            r = PyRobin()
            for i, k in enumerate(keys):
                r.insert(tuple(int(v) for v in k), i)
            if rc != 0:          # refused: only where the restatement met a probe distance at the limit
                assert rc == sage.ERR_CAPACITY and r.max_dist >= 127, (ci, n, r.max_dist)
                continue
            assert r.max_dist < 128, (ci, n, r.max_dist)
            assert list(out[:m]) == r.order(), (ci, n)


def test_product_replay_refuses_what_it_does_not_model(sage):
    """beyond a probe distance of 128 tsl::robin_map forces a growth the replay does not model: an
    error, not a wrong order (voxels that all hash to one bucket)"""
    L = sage.lib()
    rng = np.random.default_rng(29)
    keys = np.ascontiguousarray(np.array(_voxels_with_hash(lambda h: h == 0x12345, 200, rng, span=3000), dtype=np.int32))
    out = np.zeros(len(keys), dtype=np.uint32)
    assert L.sageicp_robin_iteration_order(keys.ctypes.data, len(keys), out.ctypes.data) == sage.ERR_CAPACITY
    assert b"probe distance" in L.sageicp_last_error()
    assert L.sageicp_robin_iteration_order(keys.ctypes.data, 100, out.ctypes.data) == 0      # 100 in a row is fine


@pytest.mark.parametrize("seed,n,spread,p_far", [(1, 300, 6, 0.5), (2, 5000, 20, 0.1), (3, 5000, 12, 0.6), (4, 40000, 40, 0.03),
                                                 (5, 2000, 8, 1.0), (6, 7, 2, 0.5), (7, 20000, 25, 0.3)])
def test_listed_sweep_equals_the_sweep_as_written(sage, seed, n, spread, p_far):
    """a map whose points live in HBM knows only WHICH voxels are far (the device finds them); its host-side bucket array
    must then erase — and skip — exactly what the reference's erase-while-iterating sweep does (VoxelHashMap.cpp:176-184):
    RobinTable::sweep_erase_listed against sweep_erase, erasure order and the order of what is left"""
    rng = np.random.default_rng(seed)
    vox = np.unique(rng.integers(-spread, spread + 1, size=(3 * n, 3)), axis=0)
    rng.shuffle(vox)
    vox = vox[:n]
    far = (rng.random(len(vox)) < p_far).astype(np.uint8)
    # clustered far sets as a real sweep sees them: everything beyond a radius
    if seed % 2 == 0:
        far = (np.linalg.norm(vox, axis=1) > 0.7 * spread).astype(np.uint8)
    e0, a0 = sage.robin_sweep(vox, far, listed=False)
    e1, a1 = sage.robin_sweep(vox, far, listed=True)
    assert np.array_equal(e0, e1) and np.array_equal(a0, a1)
    assert len(e0) + len(a0) == len(vox) and set(e0.tolist()) <= set(np.flatnonzero(far).tolist())
    if far.sum() > 50 and far.mean() > 0.2:
        assert len(e0) < far.sum()           # (the sweep does skip some: the behaviour under test exists)


def test_listed_sweep_on_small_tables_whose_runs_wrap_around_the_array(sage):
    """the bucket array is a ring: a run of entries that wraps around its end lets a shift carry an entry the sweep had
    skipped from bucket 0 to the last bucket, ahead of the loop again, where the sweep as written erases it after all.
    Small, half-far tables meet that a few times in a thousand (found on the GPU by the device-update property test:
    one voxel too many survived); seeds 122, 896 and 2196 of this generator are such tables"""
    wrapped = 0
    for seed in list(range(3000)):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(2, 120))
        spread = int(rng.choice([3, 10, 30]))
        vox = np.unique(rng.integers(-spread, spread + 1, size=(3 * n, 3)), axis=0)
        rng.shuffle(vox)
        vox = vox[:n]
        far = (rng.random(len(vox)) < rng.choice([0.2, 0.5, 0.8])).astype(np.uint8)
        e0, a0 = sage.robin_sweep(vox, far, listed=False)
        e1, a1 = sage.robin_sweep(vox, far, listed=True)
        assert np.array_equal(e0, e1) and np.array_equal(a0, a1), "seed %d" % seed
        # a survivor that is far and stands LAST in the array came around the ring
        wrapped += int(len(a0) > 0 and far[a0[-1]] != 0)
    assert wrapped > 0
