"""N > 1 path on CPU: two gloo ranks run the query-sharded ICP loop (contiguous shards, replicated
map, all-reduce of the Gauss-Newton sums, identical solve on every rank) with the CPU oracle as
the per-shard compute, and must reproduce the unsharded registration."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import oracle
    import sage_icp_amd as sage          # host-side map + generator only (no device needed)
    from sage_icp_amd import synthetic as syn
    from sage_icp_amd.sharding import shard_bounds

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    w = syn.make_workload("c2", lambda: sage.VoxelHashMap(1.0, 100.0), scale=0.02)
    om = oracle.Map(1.0, 100.0)
    om.add_points(w["stream"])
    p = syn.PARAMS["cold"]
    scan = w["scan"]
    lo, hi = shard_bounds(len(scan), rank, world)
    init = oracle.IDENTITY
    source = oracle.transform_points(init, scan[lo:hi])
    T_icp = oracle.IDENTITY.copy()
    iters = 0
    for _ in range(500):
        src, tgt = om.get_correspondences(source, p["max_dist"], p["sem_th"], nthreads=1)
        _, JTJ, JTr = oracle.align_clouds(src, tgt, p["kernel"], nthreads=1)
        buf = torch.from_numpy(np.concatenate([JTJ.ravel(), JTr, [float(len(src))]]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)         # the one exchange step per iteration
        b = buf.numpy()
        x = oracle.ldlt_solve6(b[:36].reshape(6, 6), -b[36:42])
        est = oracle.se3_exp(x)
        source = oracle.transform_points(est, source)
        T_icp = oracle.se3_mul(est, T_icp)
        iters += 1
        if np.linalg.norm(oracle.se3_log(est)) < 1e-4:
            break
    pose = oracle.se3_mul(T_icp, init)
    ref = None
    if rank == 0:
        ref, st = om.register_frame(scan, init, p["max_dist"], p["kernel"], p["sem_th"], nthreads=1)
        ref = np.append(ref, st.iterations)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.append(pose, iters))
    if ref is not None:
        np.save(os.path.join(out_dir, "ref.npy"), ref)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_partition():
    from sage_icp_amd.sharding import shard_bounds
    for n in (0, 1, 7, 8, 9, 120000, 500001):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(s for s in sizes if s or n < world) <= -(-n // world)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_icp_matches_unsharded(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    ref = np.load(tmp_path / "ref.npy")
    assert np.array_equal(r0, r1), "ranks must take identical decisions on identical reduced sums"
    assert r0[7] == ref[7], "same iteration count as the unsharded loop"
    assert np.allclose(r0[:7], ref[:7], rtol=0, atol=1e-9)
