#!/usr/bin/env python
"""K8 (query_flags_kernel) alone on one launch over every point of a big tree — bench.py's config-5 query leg without the rest.
usage: [PCV_HIP_LIBRARY=...] python tools/k8_probe.py [points]   -> GB/s of encoded bytes + flags, ms per launch (HIP events)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import point_cloud_viewer_amd as pcv
from bench import make_cloud

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
dev = torch.device("cuda", 0)
x, y, z, rgb = make_cloud(torch, n, 5, dev)
ctx = pcv.Context(0)
t = ctx.build(0.001, None, x, y, z, rgb)
del x, y, z, rgb
torch.cuda.empty_cache()
meta = t.meta()
lo, hi = np.asarray(meta["bbox_min"]), np.asarray(meta["bbox_max"])
big = ctx.shapes([("aabb", lo - 1.0, hi - (hi - lo) * 0.05)])
bpc = {1: 1, 2: 2, 3: 4, 4: 8}
tested = enc_bytes = 0
for i in t.nodes_in_location(big)[0]:
    nd = t.node(int(i))
    tested += nd.num_points
    enc_bytes += nd.num_points * 3 * bpc[int(nd.encoding)]
ctx.set_profiling(True)
t.query_points(big, 0, capacity=1)
ms = []
for _ in range(7):
    ctx.reset_kernel_stats()
    r = t.query_points(big, 0, capacity=1)
    ms.append(ctx.kernel_stats()["cull_points_kernel"][1])
ms.sort()
print(json.dumps({"lib": os.environ.get("PCV_HIP_LIBRARY", "default"), "points": n, "tested": int(tested), "kept": int(r["count"]),
                  "ms_min_med": [round(ms[0], 4), round(ms[len(ms) // 2], 4)],
                  "GBps_med": round((enc_bytes + tested) / (ms[len(ms) // 2] * 1e-3) / 1e9, 1)}))
