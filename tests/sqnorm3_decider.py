"""The scene that settles SAGE_SQNORM3_ORDER in one command wherever the reference can be built:
ONE query and, per case, TWO map points of another label at offsets that are rotations of one another — (a, b, c) and (c, a, b) —
so that their squared distances are the same three squares added in different associations:
    order 2, (x^2 + y^2) + z^2:   d(p1) = (a^2 + b^2) + c^2 = X      d(p2) = (c^2 + a^2) + b^2 = Y
    order 0, x^2 + (y^2 + z^2):   d(p1) = a^2 + (b^2 + c^2) = Z      d(p2) = c^2 + (a^2 + b^2) = X
with (a, b, c) searched (seeded) such that X < Y and X < Z in fp64: under order 2 the strict `<` of
VoxelHashMap.cpp:89 keeps p1, under order 0 it takes p2 — whichever the nearest neighbour the reference returns, one of
the two builds of the oracle / product is the reference's.  Offsets are multiples of 2^-40 below 1/4 and the query sits on a
multiple of 2^-1 below 64 (46 bits in all), so every difference `neighbor - point` is exact and nothing but the association differs."""
import numpy as np


def decider_cases(n_cases=8, seed=2024):
    rng = np.random.default_rng(seed)
    cases = []
    while len(cases) < n_cases:
        a, b, c = (rng.integers(1 << 34, 1 << 38, size=3).astype(np.float64)) * 2.0 ** -40       # in (0.0156, 0.25), 38 bits: the squares round
        aa, bb, cc = a * a, b * b, c * c
        X, Y, Z = (aa + bb) + cc, (cc + aa) + bb, aa + (bb + cc)
        if X < Y and X < Z:
            cases.append((a, b, c))
    return cases


def decider_scene(voxel=1.0):
    """(map points (2k, 4) in insertion order, queries (k, 4), index of the map point each association picks:
    picks[2] (order 2) and picks[0] (order 0))"""
    cases = decider_cases()
    pts, qs, pick2, pick0 = [], [], [], []
    for k, (a, b, c) in enumerate(cases):
        q = np.array([10.5 + 3.0 * k, 0.5, 0.5, 40.0])              # one voxel per case, three voxels apart
        p1 = q + [a, b, c, 10.0]                                   # (label 50 against the query's 40: the distances are compared unscaled)
        p2 = q + [c, a, b, 10.0]
        pts += [p1, p2]                                            # p1 first: a tie would keep it under either order
        qs.append(q)
        pick2.append(2 * k)
        pick0.append(2 * k + 1)
    return np.array(pts), np.array(qs), {2: np.array(pick2), 0: np.array(pick0)}
