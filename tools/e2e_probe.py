#!/usr/bin/env python
"""Host arrays -> octree directory on tmpfs with different writer-thread counts (PCV_WRITER_THREADS): where the file
side of the end-to-end path saturates. usage (GPU box): python tools/e2e_probe.py [points]"""
import os
os.environ.setdefault("PCV_HIP_LIBRARY", "exp")  # the switches live in the experiment build
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import numpy as np
    import torch
    import point_cloud_viewer_amd as pcv
    from bench import make_cloud
    n = int(sys.argv[2])
    x, y, z, rgb = make_cloud(torch, n, 1, torch.device("cuda", 0))
    hx, hy, hz, hrgb = x.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy(), rgb.cpu().numpy()
    del x, y, z, rgb
    ctx = pcv.Context(0)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="pcv_e2e_probe_", dir=base)
    best = None
    for rep in range(3):
        shutil.rmtree(os.path.join(d, "o"), ignore_errors=True)
        t0 = time.perf_counter()
        t = ctx.build(0.001, None, hx, hy, hz, hrgb)
        t1 = time.perf_counter()
        t.write_dir(os.path.join(d, "o"))
        t2 = time.perf_counter()
        t.free()
        cur = (t2 - t0, t1 - t0, t2 - t1)
        best = cur if best is None or cur[0] < best[0] else best
    shutil.rmtree(d, ignore_errors=True)
    print(f"threads={os.environ.get('PCV_WRITER_THREADS', 'default')} total_ms={best[0] * 1e3:.1f} h2d_build_ms={best[1] * 1e3:.1f} "
          f"download_and_write_ms={best[2] * 1e3:.1f} Mpts/s={n / best[0] / 1e6:.1f}")
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "100000000"
    for th in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("2", "4", "6", "8", "12", "32")):
        env = dict(os.environ, PCV_WRITER_THREADS=th)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=env)
