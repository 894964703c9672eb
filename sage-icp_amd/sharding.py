"""Query sharding for multi-GPU registration (one process per GPU).

The scan shards naturally over queries: the search is independent per query
(reference core/VoxelHashMap.cpp:98-117) and the Gauss-Newton sums are associative
(core/Registration.cpp:49-53,90).  The map is replicated; each rank takes one contiguous block of
the frame (contiguous keeps GetCorrespondences' query order when rank outputs are concatenated in
rank order); per iteration the 17 partial sums are all-reduced and every rank runs the identical
solve on identical data, so all ranks take the same decisions without a broadcast.
"""


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's block: contiguous blocks of ceil(n / world) queries."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)
