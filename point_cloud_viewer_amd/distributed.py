"""Multi-GPU octree build: shard by root octant, ONE all-to-all(v), independent subtree builds.

The reference is single-process (rayon tasks over nodes, src/octree/generation.rs:152-193); subtrees below the
root are independent, which is what this module exploits across the GPUs of one node:

  1. every rank holds a contiguous slice of the input (rank order == input order);
  2. the level-1 digit c1 = ChildIndex::from_bounding_cube(root cube, p) (src/octree/node.rs:34-42) depends only on
     the point and the global root cube, so each rank computes it locally (HIP, K2 with one level);
  3. octants are owned in contiguous ranges: owner(c) = c * world // 8; a stable partition by owner followed by one
     grouped send/recv round (RCCL: a single ncclGroup == one all-to-all(v) over xGMI; planes x, y, z, rgb[, intensity])
     routes every point to its owner. Receivers concatenate in source-rank order, so each octant's stream stays in
     global input order — the property that makes per-node point order identical to the reference;
  4. each rank runs the ordinary single-GPU build on what it received, with the GLOBAL bounding box. Its local
     root holds exactly the points its octants promote to the root, in child order;
  5. the global root is the concatenation of the local roots in rank order (== child order); all other nodes are
     disjoint between ranks. No other collective touches point data.

The backend is pluggable so that the routing/merge logic can be exercised on CPU tensors with the gloo backend
(tests/test_distributed_cpu.py injects a host backend); the product backend is HIP (`HipBackend`).
"""
import numpy as np

from . import octree as _oct


def owner_of_octant(c, world):
    """Contiguous octant ranges per rank (world in 1, 2, 4, 8; other sizes leave some ranks idle)."""
    return (c * world) // 8


class HipBackend:
    """Device work for the sharded build, all through the C ABI (no CPU fallback)."""

    def __init__(self, ctx, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.device = device

    def aabb(self, x, y, z):
        return self.ctx.aabb_reduce(x, y, z)

    def owners(self, resolution, bbox, x, y, z, world):
        """(owner per point, points per owner): one HIP kernel (root octant compare + wave-aggregated counts)."""
        return self.ctx.root_owners(resolution, bbox, x, y, z, world)

    def stable_order(self, owner):
        """Indices that sort `owner` (small ints) stably: one 3-bit pass of the HIP radix sort."""
        torch = self.torch
        n = owner.numel()
        keys = owner.to(torch.int32).clone()
        idx = torch.arange(n, dtype=torch.int32, device=owner.device)
        self.ctx.sort_pairs32(keys, idx, 0, 3)
        return idx

    def build(self, resolution, bbox, x, y, z, rgb, intensity, max_points_per_node=0):
        return self.ctx.build(resolution, bbox, x, y, z, rgb, intensity, max_points_per_node)


class ShardedOctree:
    """Result of a sharded build: this rank's subtrees plus its share of the root."""

    def __init__(self, builder, local_tree, exchange_ms, counts):
        self.builder = builder
        self.local = local_tree
        self.exchange_ms = exchange_ms
        self.counts = counts  # world x world matrix: counts[src][dst]

    @property
    def num_nodes_local(self):
        return self.local.num_nodes

    @property
    def num_points_local(self):
        return self.local.num_points

    @property
    def stage_ms(self):
        ms = dict(self.local.stage_ms()) if hasattr(self.local, "stage_ms") else {}
        ms["exchange"] = self.exchange_ms
        return ms

    def free(self):
        if hasattr(self.local, "free"):
            self.local.free()

    def gather(self, dst=0):
        """Merge all ranks' node dictionaries on `dst` (tests, directory writing). Root = concatenation in rank
        order; everything else is disjoint."""
        dist = self.builder.dist
        local = self.local.to_dict()
        gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object(local, gathered, dst=dst)
        if dist.get_rank() != dst:
            return None
        merged = {}
        for part in gathered:
            for name, nd in part.items():
                if name == "r" and "r" in merged:
                    r = merged["r"]
                    r["num_points"] += nd["num_points"]
                    for f in ("xyz", "rgb", "intensity"):
                        r[f] = r[f] + nd[f]
                else:
                    assert name not in merged, f"node {name} built by two ranks"
                    merged[name] = dict(nd)
        return merged


class ShardedOctreeBuilder:
    def __init__(self, ctx, dist, device, backend=None):
        import torch
        self.torch = torch
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.backend = backend or HipBackend(ctx, device)

    # -- global bounding box (== find_bounding_box over the whole input, generation.rs:256-270) --
    def global_bbox(self, x, y, z):
        torch, dist = self.torch, self.dist
        bmin, bmax = self.backend.aabb(x, y, z)
        n_local = x.numel() if hasattr(x, "numel") else len(x)
        big = np.finfo(np.float64).max
        if n_local == 0:  # an empty slice must not pull the box towards Aabb::zero()
            bmin, bmax = np.full(3, big), np.full(3, -big)
        lo = torch.tensor(np.asarray(bmin), dtype=torch.float64, device=self.device)
        hi = torch.tensor(np.asarray(bmax), dtype=torch.float64, device=self.device)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return _oct.Aabb(lo.cpu().numpy(), hi.cpu().numpy())

    def _exchange(self, planes, order, send_counts):
        """One grouped send/recv round for all planes. planes: list of tensors with leading dim n (local input order);
        order: stable permutation grouping rows by destination; send_counts[d] rows go to rank d."""
        torch, dist = self.torch, self.dist
        world, rank = self.world, self.rank
        counts = torch.tensor(send_counts, dtype=torch.int64, device=self.device)
        allc = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(allc, counts)
        matrix = torch.stack(allc).cpu().numpy()  # matrix[src][dst]
        recv_counts = matrix[:, rank]
        n_recv = int(recv_counts.sum())
        send_off = np.concatenate([[0], np.cumsum(send_counts)])
        recv_off = np.concatenate([[0], np.cumsum(recv_counts)])
        outs, ops = [], []
        for p in planes:
            sorted_p = p.index_select(0, order)  # rows grouped by destination, input order inside each group
            out = torch.empty((n_recv,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
            outs.append(out)
            # own rows never leave the device
            out[recv_off[rank]:recv_off[rank + 1]].copy_(sorted_p[send_off[rank]:send_off[rank + 1]])
            for peer in range(world):
                if peer == rank:
                    continue
                if send_counts[peer] > 0:
                    ops.append(dist.P2POp(dist.isend, sorted_p[send_off[peer]:send_off[peer + 1]], peer))
                if recv_counts[peer] > 0:
                    ops.append(dist.P2POp(dist.irecv, out[recv_off[peer]:recv_off[peer + 1]], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):  # NCCL/RCCL: one group == one all-to-all(v)
                w.wait()
        return outs, matrix

    def build(self, resolution, bbox, x, y, z, rgb, intensity=None, max_points_per_node=0):
        torch = self.torch
        world = self.world
        timed = self.device.type == "cuda" if hasattr(self.device, "type") else False
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        owner, send_counts = self.backend.owners(resolution, bbox, x, y, z, world)
        order = self.backend.stable_order(owner)
        planes = [x, y, z, rgb] + ([intensity] if intensity is not None else [])
        outs, matrix = self._exchange(planes, order, send_counts)
        exchange_ms = 0.0
        if timed:
            e1.record()
            e1.synchronize()
            exchange_ms = e0.elapsed_time(e1)
        rx, ry, rz, rrgb = outs[:4]
        rint = outs[4] if intensity is not None else None
        tree = self.backend.build(resolution, bbox, rx, ry, rz, rrgb, rint, max_points_per_node)
        return ShardedOctree(self, tree, exchange_ms, matrix)
