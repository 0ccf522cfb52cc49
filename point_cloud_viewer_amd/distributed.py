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

    def partition(self, owner, x, y, z, rgb, intensity, dsts):
        """Stable partition of the planes by owner straight into the destination views (count / scan / scatter)."""
        self.ctx.partition_by_owner(owner, x, y, z, rgb, intensity, dsts)

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

    def _route(self, owner, send_counts, x, y, z, rgb, intensity):
        """Partition by owner and exchange: rows for rank r go to a send buffer, own rows directly into the receive
        buffer; ONE grouped send/recv round moves everything else (RCCL: one ncclGroup == one all-to-all(v))."""
        torch, dist = self.torch, self.dist
        world, rank = self.world, self.rank
        counts = torch.tensor(send_counts, dtype=torch.int64, device=self.device)
        allc = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(allc, counts)
        matrix = torch.stack(allc).cpu().numpy()  # matrix[src][dst]
        recv_counts = matrix[:, rank]
        n_recv = int(recv_counts.sum())
        n_local = int(sum(send_counts))
        send_off = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        recv_off = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        planes = {"x": x, "y": y, "z": z, "color": rgb}
        if intensity is not None:
            planes["intensity"] = intensity

        def empty_like_rows(p, rows):
            return torch.empty((rows,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)

        send = {k: empty_like_rows(p, n_local) for k, p in planes.items()}
        recv = {k: empty_like_rows(p, n_recv) for k, p in planes.items()}
        dsts = []
        for r in range(world):
            buf, off, cnt = (recv, recv_off[rank], send_counts[rank]) if r == rank else (send, send_off[r], send_counts[r])
            d = {k: v[off:off + cnt] for k, v in buf.items()}
            d.setdefault("intensity", None)
            dsts.append(d)
        self.backend.partition(owner, x, y, z, rgb, intensity, dsts)
        ops = []
        for k in planes:
            for peer in range(world):
                if peer == rank:
                    continue
                if send_counts[peer] > 0:
                    ops.append(dist.P2POp(dist.isend, send[k][send_off[peer]:send_off[peer + 1]], peer))
                if recv_counts[peer] > 0:
                    ops.append(dist.P2POp(dist.irecv, recv[k][recv_off[peer]:recv_off[peer + 1]], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return recv, matrix

    def build(self, resolution, bbox, x, y, z, rgb, intensity=None, max_points_per_node=0):
        torch = self.torch
        world = self.world
        timed = self.device.type == "cuda" if hasattr(self.device, "type") else False
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        owner, send_counts = self.backend.owners(resolution, bbox, x, y, z, world)
        recv, matrix = self._route(owner, send_counts, x, y, z, rgb, intensity)
        exchange_ms = 0.0
        if timed:
            e1.record()
            e1.synchronize()
            exchange_ms = e0.elapsed_time(e1)
        tree = self.backend.build(resolution, bbox, recv["x"], recv["y"], recv["z"], recv["color"], recv.get("intensity"),
                                  max_points_per_node)
        return ShardedOctree(self, tree, exchange_ms, matrix)
