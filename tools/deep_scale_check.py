#!/usr/bin/env python
"""Ad-hoc check of the deep-tree path at scale (run on the GPU box): 10 M points in an 8 km cube at 1 mm with 2.1 M
duplicates -> 23 levels with the reference's capacity; every node against the closed-form oracle."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import synthetic
rng = np.random.default_rng(5)
n_bg = 8_000_000
x = np.concatenate([rng.normal(4000, 300, n_bg), np.full(1_500_000, 4321.0009), np.full(600_000, 100.5)])
y = np.concatenate([rng.normal(4000, 300, n_bg), np.full(1_500_000, 3999.25), np.full(600_000, 8000.125)])
z = np.concatenate([rng.normal(4000, 300, n_bg), np.full(1_500_000, 4100.5), np.full(600_000, 20.0)])
perm = rng.permutation(x.size); x, y, z = x[perm], y[perm], z[perm]
rgb = synthetic.hash_colors(x.size)
lo, hi = np.zeros(3), np.full(3, 8192.0)
t0 = time.time(); want = O.build_closed(0.001, lo, hi, x, y, z, rgb, threads=64); t1 = time.time()
ctx = pcv.Context(0)
t = ctx.build(0.001, pcv.Aabb(lo, hi), x, y, z, rgb); t2 = time.time()
t = ctx.build(0.001, pcv.Aabb(lo, hi), x, y, z, rgb); t3 = time.time()
print("oracle s", round(t1 - t0, 2), "gpu build (host inputs) s", round(t3 - t2, 3), "nodes", t.num_nodes, "info", t.build_info(),
      "max level", max(v["level"] for v in want.nodes.values()), "stages", {k: round(v, 2) for k, v in t.stage_ms().items()})
got = t.to_dict()
assert set(got) == set(want.nodes)
bad = [k for k, v in want.nodes.items() if got[k]["xyz"] != v["xyz"] or got[k]["rgb"] != v["rgb"] or got[k]["num_points"] != v["num_points"]]
print("mismatching nodes:", len(bad))
