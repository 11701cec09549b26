"""The N > 1 path on CPU: block-cyclic sharding of the image plane over ranks and the single
gather that reassembles it, with gloo and world_size 2 (and 4).  The per-rank "render" is the
oracle (allowed in tests), addressed through the same gr_range index map the HIP kernels use."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, tmp, use_async=False):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gradus_jl_amd as G
    from oracle import oracle as O

    plan = G.shard_plan(W, H, world, rank, block_cols=4)
    rg = plan.ray_range()
    assert rg.count == W * H // world and rg.stride_blocks == world
    # this rank's rays, in local order, through the C-ABI index map
    j = np.arange(plan.count)
    gi = np.array([plan.global_index(int(k)) for k in j])
    assert len(np.unique(gi)) == plan.count and gi.max() < W * H
    x = np.array([0.0, 100.0, math.radians(85), 0.0])
    cfg = O.make_config("kerr", (1.0, 0.0), disc=(0.0, 40.0), lambda_max=200.0)
    v = np.concatenate([O.render_velocities(cfg, x, (-9.5, 9.5), (-9.5, 9.5), W, H, i0=int(i), n=1) for i in gi])
    pts = O.trace(cfg, x, v, nthreads=1)
    local = torch.from_numpy(O.apply_pf(cfg, pts, 200.0, pf_id=O.PF_AFFINE_TIME, filter_id=O.FILTER_EARLY_TERM,
                                        nthreads=1))
    if use_async:
        # the pipelined form bench.py uses: start the gather, do other work, then collect
        # (use_async == 2: with the single-tensor receive buffers bench.py hands in -- no stacking copy)
        recv = G.gather_buffers(plan, local) if (use_async == 2 and rank == 0) else None
        handle = G.gather_image_async(local, plan, recv_bufs=recv)
        _ = local.sum()
        img = handle.result()
    else:
        img = G.gather_image(local, plan)
    if rank == 0:
        np.save(os.path.join(tmp, "img.npy"), img.numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,use_async", [(2, False), (4, False), (2, True), (4, 2), (8, 2)])
def test_sharded_render_equals_single(oracle, G, tmp_path, world, use_async):
    # world 8 = the node the driver scales to (1 -> 8 MI355X): the same deal, gather and reassembly on gloo
    W = H = 16 if world < 8 else 32
    mp.spawn(_worker, args=(world, _free_port(), W, H, str(tmp_path), use_async), nprocs=world, join=True)
    img = np.load(tmp_path / "img.npy")
    x = np.array([0.0, 100.0, math.radians(85), 0.0])
    cfg = oracle.make_config("kerr", (1.0, 0.0), disc=(0.0, 40.0), lambda_max=200.0)
    ref = oracle.rendergeodesics(cfg, x, (-9.5, 9.5), (-9.5, 9.5), W, H)
    assert img.shape == (H, W)
    np.testing.assert_array_equal(np.isnan(img), np.isnan(ref))
    np.testing.assert_array_equal(img[~np.isnan(img)], ref[~np.isnan(ref)])


def _points_worker(rank, world, port, W, H, tmp):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gradus_jl_amd as G
    from oracle import oracle as O

    plan = G.shard_plan(W, H, world, rank)
    gi = np.array([plan.global_index(int(k)) for k in range(plan.count)])
    x = np.array([0.0, 100.0, math.radians(85), 0.0])
    cfg = O.make_config("kerr", (1.0, 0.0), disc=(0.0, 40.0), lambda_max=200.0)
    v = np.concatenate([O.render_velocities(cfg, x, (-9.5, 9.5), (-9.5, 9.5), W, H, i0=int(i), n=1) for i in gi])
    pts = np.ascontiguousarray(O.trace(cfg, x, v, nthreads=1))
    assert pts.dtype.itemsize == 152
    local = torch.from_numpy(pts.view(np.uint8).copy())
    full = G.gather_points(local, plan)
    if rank == 0:
        np.save(os.path.join(tmp, "pts.npy"), full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [8])
def test_sharded_endpoints_gather_equals_single(oracle, G, tmp_path, world):
    """The one exchange of the path whose size matters (637 MB at 2048²): every rank's 152-byte records in
    local order -> ONE gather -> image order on rank 0.  World size 8 = the node the driver scales to, on gloo."""
    W, H = 16, 8
    mp.spawn(_points_worker, args=(world, _free_port(), W, H, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "pts.npy").view(oracle.POINT_DTYPE if hasattr(oracle, "POINT_DTYPE") else G._lib.POINT_DTYPE)
    x = np.array([0.0, 100.0, math.radians(85), 0.0])
    cfg = oracle.make_config("kerr", (1.0, 0.0), disc=(0.0, 40.0), lambda_max=200.0)
    v = oracle.render_velocities(cfg, x, (-9.5, 9.5), (-9.5, 9.5), W, H, i0=0, n=W * H)
    ref = oracle.trace(cfg, x, v, nthreads=2)
    assert got.size == W * H
    for f in ("status", "lambda_max", "x", "v", "x_init", "v_init"):
        np.testing.assert_array_equal(got[f], ref[f], err_msg=f)


def test_shard_plan_covers_image_exactly(G):
    for W, H, world in ((2048, 2048, 8), (2048, 2048, 4), (1024, 1024, 2), (96, 40, 3), (20, 20, 1)):
        seen = np.zeros(W * H, dtype=np.int32)
        for r in range(world):
            plan = G.shard_plan(W, H, world, r)
            rg = plan.ray_range()
            j = np.arange(rg.count)
            b = j // rg.block
            i = rg.first + b * rg.stride_blocks * rg.block + (j - b * rg.block)
            seen[i] += 1
            assert plan.global_index(int(j[-1])) == i[-1]
        assert np.all(seen == 1)
    with pytest.raises(ValueError):
        G.shard_plan(20, 20, 8, 0)


def test_ray_shards_partition_a_polar_plane():
    """distributed.ray_shard: the block-cyclic deal of a separable ray set's rays over the ranks is a partition, whole and
    ragged planes, and local -> global is the map the C ABI documents for gr_rayset.sep_first / sep_block / sep_stride."""
    import gradus_jl_amd as G
    from gradus_jl_amd.distributed import ray_shard

    for nr, nt in ((64, 48), (100, 77), (4096, 4096), (5, 9), (8, 8)):
        plane = G.PolarPlane(G.GeometricGrid(), Nr=nr, Nθ=nt)
        for world in (1, 2, 3, 8):
            shards = [ray_shard(plane, world, r) for r in range(world)]
            assert sum(s.count for s in shards) == nr * nt
            if nr * nt <= 10_000:
                seen = np.concatenate([[s.global_index(j) for j in range(s.count)] for s in shards]).astype(np.int64)
                assert np.array_equal(np.sort(seen), np.arange(nr * nt))
            for s in shards:
                if s.count:
                    j = s.count - 1
                    assert s.global_index(j) == s.first + (j // s.block) * s.stride + j % s.block < nr * nt
            if nr == 4096:
                assert shards[0].block == 8 * 4096 and max(s.count for s in shards) - min(s.count for s in shards) <= shards[0].block


def _sum_histograms(rank, world, port, out):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        part = torch.from_numpy(rng.uniform(0, 1, 180))
        total = part.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)           # the one exchange step of the sharded line profile
        out.put((rank, part.numpy(), total.numpy()))
    finally:
        dist.destroy_process_group()


def test_histogram_all_reduce_two_ranks():
    """The collective of distributed.lineprofile_sharded on gloo: every rank ends with the sum of the partial histograms."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sum_histograms, args=(r, 2, 29611, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    parts = {r: a for r, a, _ in got}
    for _, _, total in got:
        np.testing.assert_allclose(total, parts[0] + parts[1], rtol=1e-15)
