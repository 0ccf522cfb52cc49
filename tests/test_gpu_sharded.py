"""HIP side of the sharded build on one GPU: (a) the real ShardedOctreeBuilder over RCCL with world_size 1;
(b) 2/4/8 *virtual* ranks executed one after the other on the same device — same routing rule, same kernels —
merged and compared with one oracle build of the whole cloud."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import distributed as pdist, synthetic

pytestmark = pytest.mark.gpu


def _same(merged, want):
    assert set(merged) == set(want.nodes)
    for name, nd in want.nodes.items():
        g = merged[name]
        for f in ("num_points", "encoding", "xyz", "rgb"):
            assert g[f] == nd[f], (name, f)


def test_world_size_one_over_rccl():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(300_000, seed=23, num_clusters=5, extent=400.0,
                                                               sigma_range=(0.05, 6.0))
        ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        b = pdist.ShardedOctreeBuilder(ctx, dist, dev)
        tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
        trgb = torch.from_numpy(rgb).cuda()
        bbox = b.global_bbox(tx, ty, tz)
        assert np.array_equal(bbox.min, bmin) and np.array_equal(bbox.max, bmax)
        r = b.build(0.001, bbox, tx, ty, tz, trgb)
        _same(r.gather(0), O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4))
        assert r.stage_ms["exchange"] >= 0
        # 5 M points: the routed input takes the single-chain build (level-1 chain state in, depth-binned chain pass),
        # once per ownership mode
        x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(5_000_000, seed=29, num_clusters=12, extent=500.0,
                                                               sigma_range=(0.3, 9.0))
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=8)
        tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
        trgb = torch.from_numpy(rgb).cuda()
        for mode in ("buckets", "octants"):
            b = pdist.ShardedOctreeBuilder(ctx, dist, dev, shard_mode=mode)
            bbox = b.global_bbox(tx, ty, tz)
            r = b.build(0.001, bbox, tx, ty, tz, trgb)
            assert r.local.build_info()["single_chain"], r.local.build_info()
            ex = r.exchange_info()
            assert ex["shard_mode"] == mode and ex["bytes_per_row"] == 16 and ex["points_owned_per_rank"] == [5_000_000]
            _same(r.gather(0), want)
            r.free()
    finally:
        dist.destroy_process_group()


def test_route_buckets_and_partition_kernels():
    import torch
    n = 300_001
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=31, num_clusters=9, extent=200.0, sigma_range=(0.02, 9.0))
    inten = (np.arange(n) % 251).astype(np.float32)
    ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bbox = pcv.Aabb(bmin, bmax)
    tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
    trgb, tint = torch.from_numpy(rgb).cuda(), torch.from_numpy(inten).cuda()
    bucket, counts, state = ctx.route_buckets(0.001, bbox, tx, ty, tz, trgb, with_state=True)
    keys = O.chain_keys64(bmin, bmax, 0.001, 2, x, y, z)
    want = ((keys >> np.uint64(57)).astype(np.int64) & 63)
    assert np.array_equal(bucket.cpu().numpy(), want)
    assert np.array_equal(counts, np.bincount(want, minlength=64))
    # the level-1 chain state that crosses the exchange: octant digit + raw Float32 codes
    o, cx, cy, cz = O.chain_state1(bmin, bmax, 0.001, x, y, z)
    c = rgb.astype(np.uint32)
    assert np.array_equal(state["oct_rgb"].cpu().numpy().view(np.uint32), o | (c[:, 0] << 8) | (c[:, 1] << 16) | (c[:, 2] << 24))
    for got, ref in ((state["cx"], cx), (state["cy"], cy), (state["cz"], cz)):
        assert np.array_equal(got.cpu().numpy().view(np.uint32), ref)
    planes = [tx, ty, tz, trgb, tint, state["oct_rgb"].view(torch.uint8)[::4].contiguous(), state["cx"]]
    for world in (1, 3, 8):
        rank_of, _ = pdist.plan_buckets(counts, world, 5000, True)
        owner = rank_of[want].astype(np.int64)
        cnt = np.bincount(owner, minlength=world)
        dsts = [[torch.empty((int(c),) + tuple(p.shape[1:]), dtype=p.dtype, device="cuda") for p in planes] for c in cnt]
        ctx.partition_by_owner(bucket, planes, dsts, rank_of)
        for d in range(world):
            sel = torch.from_numpy(owner == d).cuda()
            for p, got in zip(planes, dsts[d]):
                assert torch.equal(got, p[sel])  # stable, every plane (8-, 3-, 4- and 1-byte rows)


def test_two_pass_routing_equals_state_plus_partition():
    """pcv_route_plan + pcv_route_scatter (the level-1 state computed in the scatter pass, never stored in input order) deliver
    exactly what pcv_route_buckets(state) + pcv_partition_by_owner deliver: bucket, counts, and every owner's rows of the four
    state planes (+ intensity) in input order — checked against the oracle's level-1 state (codec.rs:102-121 Float32 arm,
    node.rs:34-42) for 1, 3 and 8 owners, an odd point count and an input that ends inside a tile."""
    import torch
    n = 300_001
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=32, num_clusters=9, extent=200.0, sigma_range=(0.02, 9.0))
    inten = (np.arange(n) % 241).astype(np.float32)
    ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bbox = pcv.Aabb(bmin, bmax)
    tx, ty, tz = (torch.from_numpy(a).cuda() for a in (x, y, z))
    trgb, tint = torch.from_numpy(rgb).cuda(), torch.from_numpy(inten).cuda()
    bucket, tile_hist, counts = ctx.route_plan(0.001, bbox, tx, ty, tz)
    keys = O.chain_keys64(bmin, bmax, 0.001, 2, x, y, z)
    want = ((keys >> np.uint64(57)).astype(np.int64) & 63)
    assert np.array_equal(bucket.cpu().numpy().astype(np.int64), want)
    assert np.array_equal(counts, np.bincount(want, minlength=64))
    th = tile_hist.cpu().numpy().astype(np.int64)
    assert th.shape == ((n + 4095) // 4096, 64)
    for t in (0, 7, th.shape[0] - 1):
        assert np.array_equal(th[t], np.bincount(want[t * 4096:(t + 1) * 4096], minlength=64))
    o, cx, cy, cz = O.chain_state1(bmin, bmax, 0.001, x, y, z)
    c = rgb.astype(np.uint32)
    ref = {"oct_rgb": o | (c[:, 0] << 8) | (c[:, 1] << 16) | (c[:, 2] << 24), "cx": cx, "cy": cy, "cz": cz}
    for world in (1, 3, 8):
        rank_of, _ = pdist.plan_buckets(counts, world, 5000, True)
        owner = rank_of[want].astype(np.int64)
        cnt = np.bincount(owner, minlength=world)
        dsts = [dict({k: torch.empty(int(q), dtype=torch.int32, device="cuda") for k in ("oct_rgb", "cx", "cy", "cz")},
                     intensity=torch.empty(int(q), dtype=torch.float32, device="cuda")) for q in cnt]
        ctx.route_scatter(0.001, bbox, tx, ty, tz, trgb, bucket, tile_hist, rank_of, dsts, intensity=tint)
        for d in range(world):
            sel = owner == d
            for k in ("oct_rgb", "cx", "cy", "cz"):
                assert np.array_equal(dsts[d][k].cpu().numpy().view(np.uint32), ref[k][sel]), (world, d, k)
            assert np.array_equal(dsts[d]["intensity"].cpu().numpy(), inten[sel])


def test_octants_only_plan_is_the_level_1_digit_of_the_full_plan():
    """PCV_ROUTE_OCTANTS_ONLY (round 6; BASELINE north_star: shard by the top-3-bit prefix): bucket = d1 << 3 with d1 the digit
    ChildIndex::from_bounding_cube gives against the root cube (node.rs:34-42) — the upper three bits of the full plan's bucket —
    for tame points, NaN / infinite coordinates, an input that ends inside a group of four, and unaligned views (the general
    kernel); the scatter pass then delivers the same rows as with the full plan when ownership goes by octant."""
    import torch
    n = 200_003
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=35, num_clusters=7, extent=150.0, sigma_range=(0.05, 8.0))
    x[5], y[77], z[4099] = np.nan, np.inf, -np.inf
    x[9000:9040] = (bmin[0] + (bmin[0] + max(bmax - bmin))) / 2.0  # on the centre plane: strict > decides (node.rs:37-41)
    ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bbox = pcv.Aabb(bmin, bmax)
    tx, ty, tz, trgb = (torch.from_numpy(a).cuda() for a in (x, y, z, rgb))
    full, _, full_counts = ctx.route_plan(0.001, bbox, tx, ty, tz)
    b8, th8, c8 = ctx.route_plan(0.001, bbox, tx, ty, tz, octants_only=True)
    want = (full.cpu().numpy() >> 3) << 3
    assert np.array_equal(b8.cpu().numpy(), want)
    assert np.array_equal(c8.reshape(8, 8)[:, 0], full_counts.reshape(8, 8).sum(axis=1)) and c8.reshape(8, 8)[:, 1:].sum() == 0
    th = th8.cpu().numpy().astype(np.int64)
    for t in (0, 3, th.shape[0] - 1):
        assert np.array_equal(th[t], np.bincount(want[t * 4096:(t + 1) * 4096], minlength=64))
    # views at odd offsets take the general kernel: the same bytes
    b8v, _, c8v = ctx.route_plan(0.001, bbox, tx[1:], ty[1:], tz[1:], octants_only=True)
    assert np.array_equal(b8v.cpu().numpy(), want[1:]) and c8v.sum() == n - 1
    for world in (2, 8):
        rank_of, _ = pdist.plan_buckets(c8, world, 5000, True, "octants")
        owner = rank_of[want].astype(np.int64)
        cnt = np.bincount(owner, minlength=world)
        out = []
        for bucket, hist in ((b8, th8), (full, ctx.route_plan(0.001, bbox, tx, ty, tz)[1])):
            dsts = [{k: torch.empty(int(q), dtype=torch.int32, device="cuda") for k in ("oct_rgb", "cx", "cy", "cz")} for q in cnt]
            ctx.route_scatter(0.001, bbox, tx, ty, tz, trgb, bucket, hist, rank_of, dsts)
            out.append(dsts)
        for d in range(world):
            for k in ("oct_rgb", "cx", "cy", "cz"):
                assert torch.equal(out[0][d][k], out[1][d][k]), (world, d, k)


@pytest.mark.parametrize("world,cap,with_intensity,compress", [(2, 0, False, True), (4, 20_000, True, True),
                                                                (8, 3_000, False, True), (4, 20_000, True, False)])
def test_virtual_ranks_on_one_gpu(world, cap, with_intensity, compress, tmp_path):
    """The real ShardedOctreeBuilder + HipBackend, N virtual ranks as threads on one GPU (tests/thread_dist.py)."""
    import torch
    from thread_dist import run_ranks
    n = 400_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=29, num_clusters=7, extent=300.0,
                                                           sigma_range=(0.02, 5.0))
    inten = (np.arange(n) % 977).astype(np.float32) if with_intensity else None
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream

    def rank_main(rank, dist):
        torch.cuda.set_device(0)
        ctx = pcv.Context(0, stream=stream)
        sl = slice(rank * n // world, (rank + 1) * n // world)
        if world == 8 and rank == 5:
            sl = slice(0, 0)  # a rank without input
        if world == 8 and rank == 4:
            sl = slice(4 * n // 8, 6 * n // 8)
        tx, ty, tz = (torch.from_numpy(np.ascontiguousarray(a[sl])).cuda() for a in (x, y, z))
        trgb = torch.from_numpy(np.ascontiguousarray(rgb[sl])).cuda()
        tint = torch.from_numpy(np.ascontiguousarray(inten[sl])).cuda() if with_intensity else None
        b = pdist.ShardedOctreeBuilder(ctx, dist, dev, compress_exchange=compress)
        bbox = b.global_bbox(tx, ty, tz)
        assert np.array_equal(bbox.min, bmin) and np.array_equal(bbox.max, bmax)
        res = b.build(0.001, bbox, tx, ty, tz, trgb, tint, max_points_per_node=cap)
        merged = res.gather(0)
        if world == 4:  # the reference's directory, written by all ranks together
            res.write_dir(str(tmp_path / "sharded"))
        return merged, res.plan

    out = run_ranks(world, rank_main)
    merged, (rank_of, split_mask) = out[0]
    with O.max_points_per_node(cap or 100_000):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    assert set(merged) == set(want.nodes)
    for name, nd in want.nodes.items():
        for f in ("num_points", "encoding", "xyz", "rgb") + (("intensity",) if with_intensity else ()):
            assert merged[name][f] == nd[f], (name, f)
    if cap:
        assert split_mask != 0 and len(set(rank_of.tolist())) == world
    if world == 4:
        with O.max_points_per_node(cap):
            O.build_literal_dir(tmp_path / "oracle", 0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
        diffs = O.compare_octrees(O.load_dir(tmp_path / "sharded"), O.load_dir(tmp_path / "oracle"))
        assert not diffs, diffs[:10]


@pytest.mark.parametrize("world,n", [(8, 3000), (3, 50_000), (5, 1)])
def test_virtual_ranks_small_clouds_and_idle_ranks(world, n):
    """Fewer work units than ranks (level-1 nodes stay leaves, some ranks receive nothing), odd world sizes, one point."""
    import torch
    from thread_dist import run_ranks
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=37, num_clusters=3, extent=40.0, sigma_range=(0.5, 3.0))
    if n == 1:
        bmin, bmax = np.array([x[0] - 1.0, y[0] - 1.0, z[0] - 1.0]), np.array([x[0] + 1.0, y[0] + 1.0, z[0] + 1.0])
    if n == 3000:  # a loose box: every point lies in the upper-x half of the root cube -> at most 4 of 8 octants in use
        bmin = bmin - np.array([3.0 * (bmax[0] - bmin[0]), 0.0, 0.0])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream

    def rank_main(rank, dist):
        torch.cuda.set_device(0)
        ctx = pcv.Context(0, stream=stream)
        sl = slice(rank * n // world, (rank + 1) * n // world)
        tx, ty, tz = (torch.from_numpy(np.ascontiguousarray(a[sl])).cuda() for a in (x, y, z))
        trgb = torch.from_numpy(np.ascontiguousarray(rgb[sl])).cuda()
        b = pdist.ShardedOctreeBuilder(ctx, dist, dev)
        res = b.build(0.001, pcv.Aabb(bmin, bmax), tx, ty, tz, trgb)
        return res.gather(0), res.num_nodes_local

    out = run_ranks(world, rank_main)
    _same(out[0][0], O.build_closed(0.001, bmin, bmax, x, y, z, rgb))
    if n == 3000:
        assert sum(1 for _, nodes in out if nodes == 0) >= 1  # at least one rank had nothing to build


def test_virtual_ranks_deep_tree():
    """More than 21 levels (two-word keys) through the sharded build: duplicates in an 8 km cube at 1 mm."""
    import torch
    from thread_dist import run_ranks
    world = 3
    rng = np.random.default_rng(23)
    x = np.concatenate([rng.uniform(0, 8192, 15_000), np.full(2600, 5000.5), np.full(2400, 5000.5 + 0.004)])
    y = np.concatenate([rng.uniform(0, 8192, 15_000), np.full(2600, 123.0), np.full(2400, 123.0)])
    z = np.concatenate([rng.uniform(0, 8192, 15_000), np.full(2600, 8000.25), np.full(2400, 8000.25)])
    perm = rng.permutation(x.size)
    x, y, z = x[perm], y[perm], z[perm]
    rgb = synthetic.hash_colors(x.size)
    n = x.size
    bmin, bmax = np.zeros(3), np.full(3, 8192.0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream

    def rank_main(rank, dist):
        torch.cuda.set_device(0)
        ctx = pcv.Context(0, stream=stream)
        sl = slice(rank * n // world, (rank + 1) * n // world)
        tx, ty, tz = (torch.from_numpy(np.ascontiguousarray(a[sl])).cuda() for a in (x, y, z))
        trgb = torch.from_numpy(np.ascontiguousarray(rgb[sl])).cuda()
        b = pdist.ShardedOctreeBuilder(ctx, dist, dev)
        return b.build(0.001, pcv.Aabb(bmin, bmax), tx, ty, tz, trgb, max_points_per_node=1000).gather(0)

    merged = run_ranks(world, rank_main)[0]
    with O.max_points_per_node(1000):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    assert max(v["level"] for v in want.nodes.values()) > 21
    _same(merged, want)
