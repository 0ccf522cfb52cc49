"""Keep the oracle's formulations in lock-step: literal (file streaming) == closed form (chain keys +
closed-form promotion) == slow pure-python re-derivation (tests/pyref.py)."""
import numpy as np
import pytest

import oracle_lib as O
import pyref
from point_cloud_viewer_amd import synthetic


def _assert_same(a, b):
    diffs = O.compare_octrees(a, b)
    assert not diffs, "\n".join(diffs[:20])


def test_literal_equals_closed_uniform_ecef():
    x, y, z, rgb, bmin, bmax = synthetic.uniform_ecef(300_000)
    a = O.build_literal(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    b = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    _assert_same(a, b)
    assert a.total_points() == 300_000 and len(a.nodes) > 1


def test_literal_equals_closed_deep_clusters_with_intensity():
    # 3 tight clusters -> deep tree through f32, u16 and u8 levels; odd batch size; intensity carried
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(400_000, seed=5, num_clusters=3, extent=300.0,
                                                           sigma_range=(0.02, 0.6))
    inten = (np.arange(x.size) % 1000).astype(np.float32) * 0.25
    a = O.build_literal(0.001, bmin, bmax, x, y, z, rgb, inten, batch_size=77_777, threads=4)
    b = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    _assert_same(a, b)
    assert max(n["level"] for n in a.nodes.values()) >= 6
    assert any(n["num_points"] > 0 and n["intensity"] for n in a.nodes.values())
    assert {n["encoding"] for n in a.nodes.values()} >= {2, 3}


def test_literal_equals_closed_small_node_capacity():
    # lowered node capacity -> hundreds of nodes, zero-point nodes, resolution-limited leaves (u8 levels)
    with O.max_points_per_node(500):
        x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(60_000, seed=9, num_clusters=5, extent=20.0,
                                                               sigma_range=(0.001, 0.5))
        # duplicates to force resolution-limited nodes that hold > capacity points
        x[:3000] = x[0]
        y[:3000] = y[0]
        z[:3000] = z[0]
        a = O.build_literal(0.01, bmin, bmax, x, y, z, rgb, threads=4)
        b = O.build_closed(0.01, bmin, bmax, x, y, z, rgb, threads=4)
    _assert_same(a, b)
    assert len(a.nodes) > 100
    assert any(n["num_points"] == 0 for n in a.nodes.values())
    assert any(n["num_points"] > 500 for n in a.nodes.values() if n["level"] > 0)  # too small to split
    assert {n["encoding"] for n in a.nodes.values()} >= {1, 2}


def test_disk_backend_equals_memory_backend(tmp_path):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(250_000, seed=3, num_clusters=2, extent=100.0,
                                                           sigma_range=(0.5, 3.0))
    a = O.build_literal(0.001, bmin, bmax, x, y, z, rgb, threads=2)
    O.build_literal_dir(tmp_path / "oct", 0.001, bmin, bmax, x, y, z, rgb, threads=2)
    b = O.load_dir(tmp_path / "oct")
    _assert_same(a, b)
    import os
    names = set(os.listdir(tmp_path / "oct"))
    assert "meta.pb" in names
    for name, n in a.nodes.items():
        assert ((name + ".xyz") in names) == (n["num_points"] > 0)  # empty files are removed


def test_empty_input():
    e = np.zeros(0)
    t = O.build_closed(0.001, np.zeros(3), np.ones(3), e, e, e, np.zeros((0, 3), np.uint8))
    u = O.build_literal(0.001, np.zeros(3), np.ones(3), e, e, e, np.zeros((0, 3), np.uint8))
    assert len(t.nodes) == 0 and len(u.nodes) == 0


@pytest.mark.parametrize("seed,res,cap", [(1, 0.001, 300), (2, 0.05, 200), (3, 1e-5, 400)])
def test_against_pure_python_rederivation(seed, res, cap):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = 4000
    centres = rng.uniform(-50, 50, (4, 3)) + np.array([1e6, -2e6, 3e6]) * (seed == 2)
    which = rng.integers(0, 4, n)
    p = rng.standard_normal((n, 3)) * rng.uniform(0.01, 3.0, 4)[which, None] + centres[which]
    rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    bmin, bmax = p.min(axis=0), p.max(axis=0)
    with O.max_points_per_node(cap):
        t = O.build_closed(res, bmin, bmax, p[:, 0], p[:, 1], p[:, 2], rgb)
        u = O.build_literal(res, bmin, bmax, p[:, 0], p[:, 1], p[:, 2], rgb)
    _assert_same(t, u)
    ref = pyref.build([tuple(r) for r in p.tolist()], [tuple(c) for c in rgb.tolist()], bmin.tolist(),
                      bmax.tolist(), res, cap)
    assert set(ref) == set(t.nodes)
    for name, (cnt, enc, xyz, col) in ref.items():
        node = t.nodes[name]
        assert node["num_points"] == cnt, name
        assert node["encoding"] == enc, name
        assert node["xyz"] == xyz, name
        assert node["rgb"] == col, name
