"""Point-query micro-benchmark: one big AABB query (63 % of the cube per axis) over the octree of a Gaussian-cluster
cloud, REPS times, then the per-kernel HIP-event averages. Run through gpurun, alone or under tools/sq_probe.sh
(PROBE_CMD="python tools/query_probe.py")."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--frac", type=float, default=0.63)
    args = ap.parse_args()
    import torch
    import bench
    import point_cloud_viewer_amd as pcv

    dev = torch.device("cuda", 0)
    x, y, z, rgb = bench.make_cloud(torch, args.points, seed=1, device=dev)
    ctx = pcv.Context(0)
    tree = ctx.build(0.001, None, x, y, z, rgb)
    del x, y, z, rgb
    meta = tree.meta()
    bmin, bmax = meta["bbox_min"], meta["bbox_max"]
    big = ctx.shapes([("aabb", bmin, bmin + (bmax - bmin) * args.frac)])
    ctx.set_profiling(True)
    tree.query_points(big, 0, capacity=1)
    ctx.reset_kernel_stats()
    for _ in range(args.reps):
        r = tree.query_points(big, 0, capacity=1)
    st = ctx.kernel_stats()
    out = {k: {"launches": v[0], "avg_ms": round(v[1] / max(v[0], 1), 4)} for k, v in st.items() if v[0]}
    out["kept"] = r["count"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
