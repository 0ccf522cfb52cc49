"""Soak: the same cloud built over and over, every octree hashed (node table + every byte of every node) and compared with
the first — a race between the library's streams, the staging buffers or the kernels' LDS phases would show as a digest
that differs now and then. Between builds: a query on the previous tree and, every few rounds, a build of another size, so
that pool blocks are recycled in changing order.
  gpurun -- 'python tools/soak.py --builds 200 --points 20000000 > gpurun_out/soak.json'"""
import argparse, json, sys, time
import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--builds", type=int, default=100)
    ap.add_argument("--points", type=int, default=20_000_000)
    ap.add_argument("--resolution", type=float, default=0.001)
    args = ap.parse_args()
    import torch
    import point_cloud_viewer_amd as pcv
    dev = torch.device("cuda:0")
    x, y, z, rgb = bench.make_cloud(torch, args.points, seed=1, device=dev)
    xs, ys, zs, rgbs = bench.make_cloud(torch, max(args.points // 7, 1_000_000), seed=3, device=dev)
    # the second cloud once more on the host, as AoS batches for the streaming ingest (every 7th round)
    pos_s = np.stack([xs.cpu().numpy(), ys.cpu().numpy(), zs.cpu().numpy()], axis=1)
    rgb_s = rgbs.cpu().numpy()
    ctx = pcv.Context(0)
    first, other_first, differing, t0 = None, None, [], time.time()
    ingests = 0
    prev = None
    for k in range(args.builds):
        t = ctx.build(args.resolution, None, x, y, z, rgb)  # device tensors, bounding box computed inside
        d = bench.digest_of_digests(bench.tree_digests(t))
        if first is None:
            first = d
        elif d != first:
            differing.append(k)
        if prev is not None:
            prev.free()
        prev = t
        if k % 3 == 0:  # a query on the tree just built (its own kernels and scratch) before the next build starts
            shapes = ctx.shapes([("aabb", [100.0, 100.0, 100.0], [600.0, 700.0, 500.0])])
            t.cull_nodes(shapes)
        if k % 5 == 4:  # another size in between
            t2 = ctx.build(args.resolution, None, xs, ys, zs, rgbs)
            d2 = bench.digest_of_digests(bench.tree_digests(t2))
            if other_first is None:
                other_first = d2
            elif d2 != other_first:
                differing.append(-k)
            t2.free()
        if k % 7 == 6:  # the same second cloud streamed through the ingest in odd-sized batches, a build in the middle of the stream
            ing = ctx.ingest(0, has_intensity=False)
            step = 300_007
            for j, at in enumerate(range(0, pos_s.shape[0], step)):
                ing.append(pos_s[at:at + step], rgb_s[at:at + step])
                if j == 1:  # another call on the context while batches are in hand
                    t.cull_nodes(ctx.shapes([("aabb", [0.0, 0.0, 0.0], [500.0, 500.0, 500.0])]))
            t3 = ing.finish(args.resolution, None)
            d3 = bench.digest_of_digests(bench.tree_digests(t3))
            ingests += 1
            if other_first is None:
                other_first = d3
            elif d3 != other_first:
                differing.append(-1000000 - k)
            t3.free()
    print(json.dumps({"builds": args.builds, "ingests": ingests, "points": args.points, "digest": first, "second_cloud_digest": other_first,
                      "differing_builds": differing, "seconds": round(time.time() - t0, 1), "ok": not differing}))


if __name__ == "__main__":
    main()
