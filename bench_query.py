#!/usr/bin/env python
"""bench_query.py — BASELINE config 4: frustum-query path on 1 GPU vs the CPU restatement.

Octree of the config-2 cloud (default 100 M points), 10 000 random camera frusta (eye uniform in the bbox, uniform
random orientation, Perspective3(aspect 1.0, fovy 1.2, near 0.1, far 100)):
  (a) F x M node relations + relative_size_on_screen (K7 cull_nodes)      -> pairs/s
  (b) Octree::get_visible_nodes for every frustum (K7b)                     -> frusta/s
  (c) point culling of node data (K8) for sampled frusta over their visible nodes -> points/s, GB/s of encoded bytes
Kernel times come from the library's HIP-event profile (events on the launch stream). The CPU leg times the
oracle (oracle/pcv_oracle_query.cpp, 1 thread) on a bounded sample. Prints one JSON line per part.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--frusta", type=int, default=10_000)
    ap.add_argument("--cull-frusta", type=int, default=100)
    ap.add_argument("--cpu-frusta", type=int, default=50)
    args = ap.parse_args()

    import numpy as np
    import torch
    import point_cloud_viewer_amd as pcv
    import oracle_lib as O
    from bench import make_cloud

    dev = torch.device("cuda", 0)
    x, y, z, rgb = make_cloud(torch, args.points, seed=1, device=dev)
    ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bmin, bmax = ctx.aabb_reduce(x, y, z)
    tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb)
    M = tree.num_nodes
    del x, y, z, rgb

    rng = np.random.default_rng(3)
    persp = O.perspective3_new(1.0, 1.2, 0.1, 100.0)
    mats = []
    for _ in range(args.frusta):
        eye = rng.uniform(bmin, bmax)
        q = rng.normal(size=4)
        q = q / math.sqrt(float(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]))
        c, _ = O.frustum_new(eye, q, persp)
        mats.append(c)
    t0 = time.perf_counter()
    shapes = ctx.shapes([("frustum", m) for m in mats])
    setup_s = time.perf_counter() - t0

    ctx.set_profiling(True)
    ctx.reset_kernel_stats()
    t0 = time.perf_counter()
    rel, sizes = tree.cull_nodes(shapes, with_sizes=True)
    wall_a = time.perf_counter() - t0
    ks = ctx.kernel_stats()["cull_nodes_kernel"]
    pairs = args.frusta * M
    print(json.dumps({"part": "cull_nodes", "frusta": args.frusta, "nodes": M, "pairs": pairs,
                      "kernel_ms": round(ks[1], 3), "pairs_per_s_kernel": round(pairs / (ks[1] * 1e-3), 1),
                      "wall_ms_incl_D2H": round(wall_a * 1e3, 1), "shape_setup_ms": round(setup_s * 1e3, 2),
                      "relation_histogram": np.bincount(rel.ravel(), minlength=3).tolist()}))

    ctx.reset_kernel_stats()
    t0 = time.perf_counter()
    vis, status = tree.visible_nodes(shapes)
    wall_b = time.perf_counter() - t0
    ks = ctx.kernel_stats()["visible_nodes_kernel"]
    nvis = np.array([len(v) for v in vis])
    print(json.dumps({"part": "visible_nodes", "frusta": args.frusta, "kernel_ms": round(ks[1], 3),
                      "frusta_per_s_kernel": round(args.frusta / (ks[1] * 1e-3), 1), "wall_ms": round(wall_b * 1e3, 1),
                      "mean_visible": float(nvis.mean()), "max_visible": int(nvis.max()),
                      "status_nonzero": int((status != 0).sum())}))

    # (c) batched point query: nodes_in_location + decode-on-the-fly culling + stable compaction, per frustum
    ctx.reset_kernel_stats()
    sample = list(range(args.cull_frusta))
    kept = 0
    t0 = time.perf_counter()
    for f in sample:
        r = tree.query_points(shapes, f, capacity=1 << 22)
        kept += r["count"]
    wall_c = time.perf_counter() - t0
    st = ctx.kernel_stats()
    kms = st["cull_points_kernel"][1] + st["query_compact_kernel"][1] + st["nodes_in_location_kernel"][1]
    print(json.dumps({"part": "query_points", "frusta": len(sample), "kept_points": kept,
                      "kernel_ms": round(kms, 3), "cull_kernel_ms": round(st["cull_points_kernel"][1], 3),
                      "frusta_per_s_kernel": round(len(sample) / (kms * 1e-3), 1),
                      "wall_ms_incl_D2H_and_python": round(wall_c * 1e3, 1)}))
    # one large location (an AABB covering a quarter of the cloud per axis) to see the streaming rate
    big = ctx.shapes([("aabb", bmin, bmin + (bmax - bmin) * 0.63)])
    ctx.reset_kernel_stats()
    r = tree.query_points(big, 0, capacity=1)
    st = ctx.kernel_stats()
    bpc = {1: 1, 2: 2, 3: 4, 4: 8}
    visited = tree.nodes_in_location(big)[0]
    tested = sum(tree.node(int(i)).num_points for i in visited)
    enc_bytes = sum(tree.node(int(i)).num_points * 3 * bpc[tree.node(int(i)).encoding] for i in visited)
    cull_ms = st["cull_points_kernel"][1]
    # K8 roofline: HBM stream of the encoded node bytes (+ 1 flag byte per point written)
    gbs = (enc_bytes + tested) / (cull_ms * 1e-3) / 1e9
    print(json.dumps({"part": "query_points_big_aabb", "nodes_visited": int(len(visited)), "points_tested": int(tested),
                      "kept_points": r["count"], "cull_kernel_ms": round(cull_ms, 3),
                      "compact_kernel_ms": round(st["query_compact_kernel"][1], 3),
                      "roofline": {"bound": "hbm", "kernel": "cull_points_kernel", "achieved": round(gbs, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                                   "algorithmic_bytes_per_launch": int(enc_bytes + tested),
                                   "points_per_s": round(tested / (cull_ms * 1e-3), 1)}}))

    # CPU restatement (1 thread) on a bounded sample
    cubes = np.array([[*tree.node(i).cube_min, tree.node(i).cube_edge] for i in range(M)])
    nodes = {pcv.node_name(tree.node(i).id_high, tree.node(i).id_low):
             dict(id=(tree.node(i).id_high, tree.node(i).id_low), num_points=tree.node(i).num_points) for i in range(M)}
    t0 = time.perf_counter()
    for f in range(args.cpu_frusta):
        O.cull_cubes(O.SHAPE_FRUSTUM, mats[f], cubes, with_sizes=True)
    cpu_a = time.perf_counter() - t0
    t0 = time.perf_counter()
    for f in range(args.cpu_frusta):
        O.get_visible_nodes(bmin, bmax, nodes, mats[f])
    cpu_b = time.perf_counter() - t0
    print(json.dumps({"part": "cpu_baseline", "kind": "port", "cores": 1, "frusta": args.cpu_frusta,
                      "cull_nodes_pairs_per_s": round(args.cpu_frusta * M / cpu_a, 1),
                      "visible_nodes_frusta_per_s": round(args.cpu_frusta / cpu_b, 2),
                      "note": "get_visible_nodes timing includes rebuilding the id->node map per call in the oracle"}))


if __name__ == "__main__":
    main()
