#!/usr/bin/env python
"""Where the streaming ingest's time goes (GPU box): host AoS batches -> pcv_ingest_append x k -> finish -> write_dir, per phase,
for several batch sizes; the one-shot SoA path beside it. usage: python tools/ingest_probe.py [points]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import point_cloud_viewer_amd as pcv
from bench import make_cloud

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
x, y, z, rgb = make_cloud(torch, n, 1, torch.device("cuda", 0))
hx, hy, hz, hrgb = x.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy(), rgb.cpu().numpy()
del x, y, z, rgb
pos = np.empty((n, 3), dtype=np.float64)
pos[:, 0], pos[:, 1], pos[:, 2] = hx, hy, hz
ctx = pcv.Context(0)
base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
d = tempfile.mkdtemp(prefix="pcv_ingest_probe_", dir=base)
try:
    for batch in [int(b) for b in os.environ.get("BATCHES", "500000,500000,1000000,4000000").split(",")]:
        rows = []
        for rep in range(4):
            shutil.rmtree(os.path.join(d, "o"), ignore_errors=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ing = ctx.ingest(n, False)
            t1 = time.perf_counter()
            for at in range(0, n, batch):
                ing.append(pos[at:at + batch], hrgb[at:at + batch])
            t2 = time.perf_counter()
            ctx.synchronize()
            t3 = time.perf_counter()
            tree = ing.finish(0.001, None)
            t4 = time.perf_counter()
            tree.write_dir(os.path.join(d, "o"))
            t5 = time.perf_counter()
            tree.free()
            rows.append([round((b - a) * 1e3, 1) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t0, t5))])
        print(f"batch {batch}: [begin, appends, drain, finish(build), write_dir, total] ms per run: {rows[1:]}  -> "
              f"{n / (min(r[5] for r in rows[1:]) * 1e-3) / 1e6:.0f} M pts/s best", flush=True)
    for rep in range(3):
        shutil.rmtree(os.path.join(d, "o"), ignore_errors=True)
        t0 = time.perf_counter()
        tree = ctx.build(0.001, None, hx, hy, hz, hrgb)
        t1 = time.perf_counter()
        tree.write_dir(os.path.join(d, "o"))
        t2 = time.perf_counter()
        tree.free()
        print(f"one-shot SoA: h2d+build {1e3 * (t1 - t0):.1f} ms, write_dir {1e3 * (t2 - t1):.1f} ms, {n / (t2 - t0) / 1e6:.0f} M pts/s", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
