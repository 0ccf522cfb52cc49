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
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_virtual_ranks_on_one_gpu(world):
    import torch
    n = 400_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=29, num_clusters=7, extent=300.0,
                                                           sigma_range=(0.02, 5.0))
    ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    backend = pdist.HipBackend(ctx, torch.device("cuda", 0))
    bbox = pcv.Aabb(bmin, bmax)
    # every virtual rank owns a contiguous input slice, computes digits + stable partition with the HIP kernels
    parts = {d: [] for d in range(world)}
    for r in range(world):
        sl = slice(r * n // world, (r + 1) * n // world)
        tx, ty, tz = (torch.from_numpy(np.ascontiguousarray(a[sl])).cuda() for a in (x, y, z))
        trgb = torch.from_numpy(np.ascontiguousarray(rgb[sl])).cuda()
        owner, counts = backend.owners(0.001, bbox, tx, ty, tz, world)
        keys = O.chain_keys64(bmin, bmax, 0.001, 1, x[sl], y[sl], z[sl])
        want_owner = ((keys >> np.uint64(60)).astype(np.int64) * world) // 8
        assert np.array_equal(owner.cpu().numpy(), want_owner)
        assert counts == np.bincount(want_owner, minlength=world).tolist()
        dsts = [dict(x=torch.empty(c, dtype=torch.float64, device="cuda"), y=torch.empty(c, dtype=torch.float64, device="cuda"),
                     z=torch.empty(c, dtype=torch.float64, device="cuda"), color=torch.empty((c, 3), dtype=torch.uint8, device="cuda"),
                     intensity=None) for c in counts]
        backend.partition(owner, tx, ty, tz, trgb, None, dsts)
        for d in range(world):
            sel = torch.from_numpy(want_owner == d).cuda()
            assert torch.equal(dsts[d]["x"], tx[sel]) and torch.equal(dsts[d]["z"], tz[sel])  # stable
            assert torch.equal(dsts[d]["color"], trgb[sel])
            parts[d].append((dsts[d]["x"], dsts[d]["y"], dsts[d]["z"], dsts[d]["color"]))
    merged = {}
    for d in range(world):  # receivers concatenate in source-rank order
        rx, ry, rz, rrgb = (torch.cat([p[i] for p in parts[d]]).contiguous() for i in range(4))
        tree = backend.build(0.001, bbox, rx, ry, rz, rrgb, None)
        for name, nd in tree.to_dict().items():
            if name == "r" and "r" in merged:
                merged["r"]["num_points"] += nd["num_points"]
                merged["r"]["xyz"] += nd["xyz"]
                merged["r"]["rgb"] += nd["rgb"]
            else:
                assert name not in merged
                merged[name] = nd
    _same(merged, O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4))
