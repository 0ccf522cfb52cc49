"""Host cost of pcv_spec_resolve (the single-chain build's critical host step) on a tree of bench size, on the CPU:
8 M Gaussian-cluster points with capacity 8 000 give about the node counts of the 100 M-point bench cloud at 100 000.
    PCV_SPEC_TIME=1 python tools/spec_resolve_time.py"""
import os
os.environ.setdefault("PCV_HIP_LIBRARY", "exp")  # the switches live in the experiment build
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PCV_SPEC_TIME", "1")

import oracle_lib as O  # noqa: E402
from point_cloud_viewer_amd import synthetic  # noqa: E402
from test_spec_cpu import _selftest  # noqa: E402

x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(8_000_000, seed=1, num_clusters=64, extent=1000.0, sigma_range=(1.0, 20.0))
ml, edges, _ = O.level_table(bmin, bmax, 0.001)
nlevels = min(ml, 21)
keys = O.chain_keys64(bmin, bmax, 0.001, nlevels, x, y, z, threads=8)
rc, prefix, *_ = _selftest(keys, 32, 8000, 0.32, 0.001, edges, nlevels)
print("status", rc, "true nodes", len(prefix))
