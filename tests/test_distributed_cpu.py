"""world_size-2 (and 4) gloo tests of the sharded build's routing / merge logic on CPU tensors.

The device work is replaced by a host backend built on the oracle (tests may use the oracle); what is under test
is point_cloud_viewer_amd.distributed: octant ownership, stable partition, the grouped send/recv exchange
(source-rank order), and the root merge. Expected result: identical to ONE oracle build over the whole cloud."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostTree:
    def __init__(self, oct_):
        self.oct = oct_
        self.num_nodes = len(oct_.nodes)
        self.num_points = oct_.total_points()

    def to_dict(self):
        return {k: dict(num_points=v["num_points"], encoding=v["encoding"], xyz=v["xyz"], rgb=v["rgb"],
                        intensity=v["intensity"], id=v["id"], level=v["level"]) for k, v in self.oct.nodes.items()}

    def stage_ms(self):
        return {}


class HostBackend:
    def __init__(self, O, cap):
        self.O, self.cap = O, cap

    def aabb(self, x, y, z):
        return self.O.aabb(x.numpy(), y.numpy(), z.numpy())

    def owners(self, resolution, bbox, x, y, z, world):
        keys = self.O.chain_keys64(bbox.min, bbox.max, resolution, 1, x.numpy(), y.numpy(), z.numpy())
        owner = torch.from_numpy(((keys >> np.uint64(60)).astype(np.int64) * world) // 8)
        return owner, torch.bincount(owner, minlength=world).tolist()

    def partition(self, owner, x, y, z, rgb, intensity, dsts):
        for k, d in enumerate(dsts):
            sel = owner == k  # boolean-mask selection keeps input order
            d["x"].copy_(x[sel])
            d["y"].copy_(y[sel])
            d["z"].copy_(z[sel])
            d["color"].copy_(rgb[sel])
            if intensity is not None:
                d["intensity"].copy_(intensity[sel])

    def build(self, resolution, bbox, x, y, z, rgb, intensity, max_points_per_node=0):
        with self.O.max_points_per_node(self.cap):
            t = self.O.build_closed(resolution, bbox.min, bbox.max, x.numpy(), y.numpy(), z.numpy(), rgb.numpy(),
                                    None if intensity is None else intensity.numpy())
        return _HostTree(t)


def _worker(rank, world, port, n, cap, with_intensity, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    import oracle_lib as O
    from point_cloud_viewer_amd import distributed as pdist, synthetic
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=17, num_clusters=6, extent=50.0,
                                                           sigma_range=(0.01, 4.0))
    inten = (np.arange(n) % 97).astype(np.float32)
    lo, hi = rank * n // world, (rank + 1) * n // world  # contiguous input slices, rank order == input order
    if rank == world - 1 and world > 2:
        lo = hi  # an empty slice on the last rank
    elif rank == world - 2 and world > 2:
        hi = n
    sl = slice(lo, hi)
    tx, ty, tz = (torch.from_numpy(np.ascontiguousarray(a[sl])) for a in (x, y, z))
    trgb = torch.from_numpy(np.ascontiguousarray(rgb[sl]))
    tint = torch.from_numpy(np.ascontiguousarray(inten[sl])) if with_intensity else None
    b = pdist.ShardedOctreeBuilder(None, dist, torch.device("cpu"), backend=HostBackend(O, cap))
    bbox = b.global_bbox(tx, ty, tz)
    assert np.array_equal(bbox.min, bmin) and np.array_equal(bbox.max, bmax)
    res = b.build(0.001, bbox, tx, ty, tz, trgb, tint)
    assert int(res.counts.sum()) == n
    merged = res.gather(dst=0)
    if rank == 0:
        with O.max_points_per_node(cap):
            want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten if with_intensity else None)
        ok = set(merged) == set(want.nodes)
        msgs = [] if ok else ["node sets differ"]
        for name, nd in want.nodes.items():
            g = merged.get(name)
            if g is None:
                continue
            for f in ("num_points", "encoding", "xyz", "rgb", "intensity"):
                if g[f] != nd[f]:
                    msgs.append(f"{name}.{f} differs")
        open(out_path, "w").write("OK" if not msgs else "\n".join(msgs[:20]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,with_intensity", [(2, False), (2, True), (4, False)])
def test_sharded_build_equals_single_build(tmp_path, world, with_intensity):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), 40_000, 700, with_intensity, str(out)), nprocs=world, join=True)
    assert out.read_text() == "OK", out.read_text()


def test_octant_ownership_is_contiguous_and_balanced():
    from point_cloud_viewer_amd.distributed import owner_of_octant
    for world in (1, 2, 4, 8):
        owners = [owner_of_octant(c, world) for c in range(8)]
        assert owners == sorted(owners) and set(owners) == set(range(world))
        assert all(owners.count(r) == 8 // world for r in range(world))
