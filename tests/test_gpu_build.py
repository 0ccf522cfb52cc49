"""End-to-end parity of the HIP octree build against the CPU oracle: node ids, counts, encodings and the exact
bytes of every node file (bit-exact: integer/byte work; the f64 chain is replayed with identical rounding)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pcv.Context(0)
    yield c
    c.close()


def assert_same(got, want, check_intensity=False):
    assert set(got) == set(want.nodes), (sorted(set(got) ^ set(want.nodes))[:10])
    for name, nd in want.nodes.items():
        g = got[name]
        assert g["id"] == nd["id"], name
        assert g["num_points"] == nd["num_points"], name
        assert g["encoding"] == nd["encoding"], name
        assert g["xyz"] == nd["xyz"], f"{name}: xyz bytes differ"
        assert g["rgb"] == nd["rgb"], f"{name}: rgb bytes differ"
        if check_intensity:
            assert g["intensity"] == nd["intensity"], f"{name}: intensity bytes differ"


def test_reference_unit_test_cloud(ctx):
    # src/octree/tests.rs:18-46
    x, y, z, rgb, bmin, bmax, res = synthetic.reference_unit_test_cloud()
    t = ctx.build(res, pcv.Aabb(bmin, bmax), x, y, z, rgb)
    got = t.to_dict()
    assert sorted(got) == ["r", "r0", "r4"]
    assert got["r"]["num_points"] == 12501 and got["r0"]["num_points"] == 0 and got["r4"]["num_points"] == 87500
    assert_same(got, O.build_closed(res, bmin, bmax, x, y, z, rgb))
    assert t.num_points == 100001


def test_uniform_ecef_loose_bbox_vs_literal_and_closed(ctx):
    x, y, z, rgb, bmin, bmax = synthetic.uniform_ecef(400_000)
    got = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb).to_dict()
    assert_same(got, O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4))
    assert_same(got, O.build_literal(0.001, bmin, bmax, x, y, z, rgb, threads=4))


def test_deep_clusters_with_intensity_and_rgba(ctx):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(500_000, seed=5, num_clusters=3, extent=300.0,
                                                           sigma_range=(0.02, 0.6))
    inten = (np.arange(x.size) % 1000).astype(np.float32) * 0.25
    rgba = np.concatenate([rgb, np.full((x.size, 1), 77, np.uint8)], axis=1)
    got = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgba, inten).to_dict()
    want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    assert_same(got, want, check_intensity=True)
    assert max(n["level"] for n in want.nodes.values()) >= 6


@pytest.mark.parametrize("cap,res,seed", [(500, 0.01, 9), (64, 0.001, 10), (2000, 1e-5, 12), (1, 0.5, 13)])
def test_small_node_capacity_many_nodes(ctx, cap, res, seed):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(60_000, seed=seed, num_clusters=5, extent=20.0,
                                                           sigma_range=(0.001, 0.5))
    x[:3000], y[:3000], z[:3000] = x[0], y[0], z[0]  # duplicates: resolution-limited nodes above capacity
    got = ctx.build(res, pcv.Aabb(bmin, bmax), x, y, z, rgb, max_points_per_node=cap).to_dict()
    with O.max_points_per_node(cap):
        want = O.build_closed(res, bmin, bmax, x, y, z, rgb, threads=4)
    assert_same(got, want)
    assert len(want.nodes) > 50


def test_float64_levels_city_scale_ecef(ctx):
    # root edge > 16.7 km at 1 mm -> Float64 root/level-1 nodes, Float32 below (codec.rs:31-40)
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(300_000, seed=6, num_clusters=8, extent=30000.0,
                                                           sigma_range=(1.0, 500.0), offset=(-2.7e6, -4.3e6, 3.8e6))
    got = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb).to_dict()
    want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    assert_same(got, want)
    assert 4 in {n["encoding"] for n in want.nodes.values()}


def test_degenerate_inputs(ctx):
    e = np.zeros(0)
    t = ctx.build(0.001, pcv.Aabb([0, 0, 0], [1, 1, 1]), e, e, e, np.zeros((0, 3), np.uint8))
    assert t.num_nodes == 0 and t.num_points == 0
    # a single point; all points identical with a zero-size bounding box (edge 0 -> NaN codes, like the reference)
    for n in (1, 7, 9, 1000):
        x = np.full(n, 2.5)
        rgb = np.arange(3 * n, dtype=np.uint8).reshape(n, 3)
        for bmin, bmax in (([0, 0, 0], [10, 10, 10]), ([2.5, 2.5, 2.5], [2.5, 2.5, 2.5])):
            want = O.build_closed(0.001, np.array(bmin, float), np.array(bmax, float), x, x, x, rgb)
            for single_chain in (None, True):  # the forced single-chain build must cope with (or hand back) the same corners
                got = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, x, x, rgb, single_chain=single_chain).to_dict()
                assert_same(got, want)


def test_special_coordinates_take_the_guarded_chain(ctx):
    """Signed zeros, denormals, huge values, NaN and infinities mixed into an ordinary cloud: the kernels route such
    points through the guarded division variant; the bytes must not change."""
    n = 60_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=41, num_clusters=4, extent=16.0, sigma_range=(0.01, 1.5))
    special = np.array([-0.0, 0.0, 5e-324, 1e-310, 2.0 ** -701, 2.0 ** -699, 1e-300, 1e200, -1e200, 2.0 ** 501,
                        float("nan"), float("inf"), -float("inf"), 8.0, -8.0, 2.0 ** -30, -(2.0 ** -30)])
    rng = np.random.default_rng(7)
    for coords in (x, y, z):
        idx = rng.choice(n, 3000, replace=False)
        coords[idx] = rng.choice(special, idx.size)
    # the second box cuts the cloud in half: the outside points collapse onto its faces (clamped codes) and the tree gets
    # deep — at 1e-7 m 28 levels, i.e. the two-word key path
    for lo, hi, resolutions in ((bmin, bmax, (0.001, 1e-7)),
                                (np.array([-8.0, -8.0, -8.0]), np.array([8.0, 8.0, 8.0]), (0.001, 1e-7)),
                                (np.array([0.0, -0.0, -8.0]), np.array([8.0, 8.0, 8.0]), (0.001, 1e-7))):
        for res in resolutions:
            with O.max_points_per_node(900):
                want = O.build_closed(res, lo, hi, x, y, z, rgb, threads=4)
            for single_chain in (None, True):
                got = ctx.build(res, pcv.Aabb(lo, hi), x, y, z, rgb, max_points_per_node=900, single_chain=single_chain).to_dict()
                assert_same(got, want)


def test_device_resident_inputs_and_computed_bbox(ctx):
    import torch
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(300_000, seed=21, num_clusters=4, extent=500.0,
                                                           sigma_range=(0.1, 5.0))
    dx, dy, dz = (torch.from_numpy(a).cuda() for a in (x, y, z))
    dc = torch.from_numpy(rgb).cuda()
    t = ctx.build(0.001, None, dx, dy, dz, dc, stage_times=True)  # bbox computed on the device (find_bounding_box)
    m = t.meta()
    assert np.array_equal(m["bbox_min"], bmin) and np.array_equal(m["bbox_max"], bmax)
    assert_same(t.to_dict(), O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4))
    ms = t.stage_ms()
    assert ms["total"] > 0 and set(ms) >= {"chain_keys", "sort_keys", "promote_encode"}


def test_written_directory_equals_oracle_directory(ctx, tmp_path):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(250_000, seed=3, num_clusters=2, extent=100.0,
                                                           sigma_range=(0.5, 3.0))
    inten = np.linspace(0, 1, x.size, dtype=np.float32)
    pcv.build_octree(tmp_path / "gpu", 0.001, pcv.Aabb(bmin, bmax), dict(x=x, y=y, z=z, color=rgb, intensity=inten),
                     attributes=("color", "intensity"), ctx=ctx)
    O.build_literal_dir(tmp_path / "cpu", 0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    a, b = O.load_dir(tmp_path / "gpu"), O.load_dir(tmp_path / "cpu")  # parses our meta.pb with the oracle's reader
    assert not O.compare_octrees(a, b)
    assert sorted(os.listdir(tmp_path / "gpu")) == sorted(os.listdir(tmp_path / "cpu"))  # same file set
    for f in os.listdir(tmp_path / "cpu"):
        if f != "meta.pb":  # meta node order is nondeterministic in the reference
            assert open(tmp_path / "gpu" / f, "rb").read() == open(tmp_path / "cpu" / f, "rb").read(), f


def test_depth_overflow_is_reported(ctx):
    # > capacity identical points and a resolution so fine that even 40 levels (all the reference's NodeId can name)
    # cannot bring the edge down to it
    n = 300
    x = np.full(n, 0.123456789)
    rgb = np.zeros((n, 3), np.uint8)
    with pytest.raises(pcv.PcvError) as ei:
        ctx.build(1e-13, pcv.Aabb([0, 0, 0], [1, 1, 1]), x, x, x, rgb, max_points_per_node=100)
    assert ei.value.code == -5  # PCV_E_DEPTH


def test_deep_trees_beyond_21_levels(ctx, tmp_path):
    """Heavy duplicates in a cube with edge / resolution > 2^21: the tree needs more levels than one 63-bit key word
    holds (second key word for levels 22..40). Node ids, counts and bytes against the oracle."""
    rng = np.random.default_rng(19)
    # (a) 8 km cube at 1 mm: 23 levels; two piles of duplicates 3 mm apart plus background
    n_bg = 20_000
    x = np.concatenate([rng.uniform(0, 8192, n_bg), np.full(3000, 1234.5678), np.full(2500, 1234.5678 + 0.003)])
    y = np.concatenate([rng.uniform(0, 8192, n_bg), np.full(3000, 777.25), np.full(2500, 777.25)])
    z = np.concatenate([rng.uniform(0, 8192, n_bg), np.full(3000, 4000.125), np.full(2500, 4000.125 - 0.002)])
    perm = rng.permutation(x.size)
    x, y, z = x[perm], y[perm], z[perm]
    rgb = synthetic.hash_colors(x.size)
    inten = (np.arange(x.size) % 113).astype(np.float32)
    lo, hi = np.zeros(3), np.full(3, 8192.0)
    with O.max_points_per_node(1000):
        want = O.build_closed(0.001, lo, hi, x, y, z, rgb, inten, threads=4)
    assert max(v["level"] for v in want.nodes.values()) > 21
    t = ctx.build(0.001, pcv.Aabb(lo, hi), x, y, z, rgb, inten, max_points_per_node=1000)
    assert t.build_info()["key_levels"] > 21
    assert_same(t.to_dict(), want)
    # node ids above 2^64 through meta.pb and the file names, read back by the oracle's loader and by open_dir
    t.write_dir(str(tmp_path / "deep"))
    with O.max_points_per_node(1000):
        O.build_literal_dir(tmp_path / "deep_oracle", 0.001, lo, hi, x, y, z, rgb, inten, threads=4)
    diffs = O.compare_octrees(O.load_dir(tmp_path / "deep"), O.load_dir(tmp_path / "deep_oracle"))
    assert not diffs, diffs[:10]
    reopened = ctx.open_dir(str(tmp_path / "deep"))
    assert sorted(reopened.node_names()) == sorted(want.nodes)
    # (b) identical points, unit cube, 1e-12: all 40 levels, then the edge is below the resolution and the leaf stays big
    n = 300
    x = np.full(n, 0.123456789)
    rgb = synthetic.hash_colors(n)
    with O.max_points_per_node(100):
        want = O.build_closed(1e-12, np.zeros(3), np.ones(3), x, x, x, rgb)
    assert max(v["level"] for v in want.nodes.values()) == 40
    assert_same(ctx.build(1e-12, pcv.Aabb([0, 0, 0], [1, 1, 1]), x, x, x, rgb, max_points_per_node=100).to_dict(), want)


def test_invalid_arguments(ctx):
    x = np.zeros(4)
    rgb = np.zeros((4, 3), np.uint8)
    with pytest.raises(pcv.PcvError):
        ctx.build(0.0, pcv.Aabb([0, 0, 0], [1, 1, 1]), x, x, x, rgb)
    with pytest.raises(pcv.PcvError):
        ctx.build(float("nan"), pcv.Aabb([0, 0, 0], [1, 1, 1]), x, x, x, rgb)
    with pytest.raises(ValueError):
        ctx.build(0.001, pcv.Aabb([0, 0, 0], [1, 1, 1]), x, x, np.zeros(3), rgb)


def test_large_build_properties(ctx):
    """5 M points: too slow for the literal oracle in a unit test, so check size-independent properties plus the
    closed-form oracle on node table level (ids/counts) and a byte-exact sample of nodes."""
    n = 5_000_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=1, num_clusters=64, extent=1000.0)
    rgb = synthetic.index_colors(n)  # colour = 24-bit index -> membership is checkable from the output
    t = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb)
    got = t.to_dict()
    assert t.num_points == n and sum(g["num_points"] for g in got.values()) == n
    # every point appears exactly once, and inside a node the input order is preserved per promoted stream
    seen = np.zeros(n, dtype=np.uint8)
    for name, g in got.items():
        c = np.frombuffer(g["rgb"], dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        idx = (c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]
        np.add.at(seen, idx, 1)
        bpc = {1: 1, 2: 2, 3: 4, 4: 8}[g["encoding"]]
        assert len(g["xyz"]) == g["num_points"] * 3 * bpc
        # each stored position decodes to within 2 quanta per climbed level of its source (cheap sanity)
    assert seen.min() == 1 and seen.max() == 1
    want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=8)
    assert_same(got, want)


def test_depth_speculation_holds_and_matches_full_depth(ctx):
    n = 4_500_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=31, num_clusters=16, extent=600.0,
                                                           sigma_range=(0.3, 10.0))
    a = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, single_chain=False)
    b = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, speculate_depth=False)
    ia, ib = a.build_info(), b.build_info()
    assert ia["attempts"] == 1 and ib["attempts"] == 1
    assert ia["key_levels"] < ib["key_levels"]  # fewer digit levels computed and sorted
    da, db = a.to_dict(), b.to_dict()
    assert set(da) == set(db)
    for k in da:
        assert da[k]["xyz"] == db[k]["xyz"] and da[k]["rgb"] == db[k]["rgb"] and da[k]["num_points"] == db[k]["num_points"]
    deepest = max(v["level"] for v in da.values())
    assert ia["key_levels"] >= deepest
    assert_same(da, O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=8))


def test_depth_speculation_failure_falls_back_to_full_depth(ctx):
    """Adversarial input order: the strided depth probe only ever sees a shallow uniform cloud, the dense cluster
    hides between the sampled indices -> the estimate is too shallow, K4 notices, the build is redone."""
    n = 1 << 22
    stride = n >> 18
    rng = np.random.default_rng(7)
    x = rng.uniform(0.0, 500.0, n)
    y = rng.uniform(0.0, 500.0, n)
    z = rng.uniform(0.0, 500.0, n)
    # visible to the exact pipeline's depth probe: every 16th point; to the single-chain build's sample: clumps of 8
    # consecutive points every 256 (every 32nd point on average)
    idx = np.arange(n)
    hidden = ((idx % stride) != 0) & ((idx % 256) >= 8)
    m = int(hidden.sum())
    x[hidden] = 250.0 + rng.normal(0.0, 0.01, m)
    y[hidden] = 125.0 + rng.normal(0.0, 0.01, m)
    z[hidden] = 333.0 + rng.normal(0.0, 0.01, m)
    rgb = synthetic.index_colors(n)
    bmin, bmax = np.zeros(3), np.full(3, 500.0)
    want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=8)
    t = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, single_chain=False)
    info = t.build_info()
    assert info["attempts"] == 2, info
    assert_same(t.to_dict(), want)
    t.free()
    # the single-chain build samples one point in 32 (in clumps) and is fooled the same way: its prediction is too shallow, the
    # exact counts notice, the exact pipeline (whose depth probe is fooled too) redoes the build
    t = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb)
    info = t.build_info()
    assert info["attempts"] == 3 and not info["single_chain"], info
    assert_same(t.to_dict(), want)


def test_build_octree_from_file(ctx, tmp_path):
    """build_octree_from_file (generation.rs:272-287): PLY (f32 coordinates + header offset) -> octree directory."""
    import struct
    x, y, z, rgb, _, _ = synthetic.gaussian_clusters(200_000, seed=41, num_clusters=3, extent=80.0, sigma_range=(0.05, 2.0))
    xf, yf, zf = x.astype(np.float32), y.astype(np.float32), z.astype(np.float32)
    inten = (np.arange(x.size) % 17).astype(np.float32)
    off = (500000.25, -4.0e6, 1234.5)
    with open(tmp_path / "cloud.ply", "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment offset: {off[0]!r} {off[1]!r} {off[2]!r}\n"
                 f"element vertex {x.size}\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                 "property uchar green\nproperty uchar blue\nproperty float intensity\nend_header\n").encode())
        rec = np.zeros(x.size, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("i", "<f4")])
        rec["x"], rec["y"], rec["z"], rec["r"], rec["g"], rec["b"], rec["i"] = xf, yf, zf, rgb[:, 0], rgb[:, 1], rgb[:, 2], inten
        f.write(rec.tobytes())
    pcv.build_octree_from_file(tmp_path / "gpu", 0.001, tmp_path / "cloud.ply", ctx=ctx)  # decode on the device
    pcv.build_octree_from_file(tmp_path / "gpu_host", 0.001, tmp_path / "cloud.ply", ctx=ctx, host_decode=True)
    px, py, pz = xf.astype(np.float64) + off[0], yf.astype(np.float64) + off[1], zf.astype(np.float64) + off[2]
    bmin, bmax = O.aabb(px, py, pz)
    O.build_literal_dir(tmp_path / "cpu", 0.001, bmin, bmax, px, py, pz, rgb, inten, threads=4)
    assert not O.compare_octrees(O.load_dir(tmp_path / "gpu"), O.load_dir(tmp_path / "cpu"))
    assert not O.compare_octrees(O.load_dir(tmp_path / "gpu_host"), O.load_dir(tmp_path / "cpu"))


def test_device_ply_decode_equals_host_decode(ctx, tmp_path):
    """pcv_build_octree_from_ply (vertex records uploaded as they are, cast + `comment offset` on the device,
    ply.rs:488-493) against the host parser + pcv_build_octree on the same file: x / y / z of different scalar types at
    unaligned offsets inside a 23-byte record, properties that are skipped, a trailing element; and the reference's own
    fixture files when the checkout is there."""
    import os
    rng = np.random.default_rng(5)
    n = 60_000
    rec = np.zeros(n, dtype=[("pad", "u1"), ("x", "<i2"), ("r", "u1"), ("y", "<f8"), ("g", "u1"), ("z", "<u4"), ("b", "u1"),
                             ("a", "u1"), ("i", "<f4")])
    rec["x"], rec["y"], rec["z"] = rng.integers(-3000, 3000, n), rng.normal(0.0, 40.0, n), rng.integers(0, 5000, n)
    rec["r"], rec["g"], rec["b"], rec["a"] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n), 7
    rec["i"] = rng.uniform(0, 100, n).astype(np.float32)
    with open(tmp_path / "mixed.ply", "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment offset: 0.125 -2000000.5 1e-3\nelement vertex {n}\n"
                 "property uchar pad\nproperty short x\nproperty uchar red\nproperty double y\nproperty uchar green\n"
                 "property uint z\nproperty uchar blue\nproperty uchar alpha\nproperty float intensity\n"
                 "element face 0\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(rec.tobytes())
    files = [(tmp_path / "mixed.ply", True)]
    for name, has_int in (("xyz_f32_rgb_u8_le.ply", False), ("xyz_f32_rgba_u8_le.ply", False), ("xyz_f32_rgb_u8_intensity_f32.ply", True)):
        if os.path.exists(os.path.join("/root/reference/src/test_data", name)):
            files.append((os.path.join("/root/reference/src/test_data", name), has_int))
    for path, has_int in files:
        dev = ctx.build_from_ply(0.01, path, with_intensity=has_int, max_points_per_node=2000)
        pts = pcv.read_ply(path)
        host = ctx.build(0.01, None, pts["x"], pts["y"], pts["z"], pts["color"], pts["intensity"] if has_int else None,
                         max_points_per_node=2000)
        assert dev.meta()["bbox_min"].tolist() == host.meta()["bbox_min"].tolist()
        a, b = dev.to_dict(), host.to_dict()
        assert set(a) == set(b) and dev.num_points == pts["x"].size
        for name in a:
            for key in ("num_points", "encoding", "xyz", "rgb"):
                assert a[name][key] == b[name][key], (path, name, key)
            ia, ib = np.frombuffer(a[name]["intensity"], np.float32), np.frombuffer(b[name]["intensity"], np.float32)
            assert np.array_equal(ia, ib, equal_nan=True), (path, name)  # the fixture's intensities are NaN
    # intensity asked for, none in the file (the reference panics); a missing file
    small = np.zeros(4, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    with open(tmp_path / "noint.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" + small.tobytes())
    with pytest.raises(pcv.PcvError, match="requested but the PLY has none"):
        ctx.build_from_ply(0.01, tmp_path / "noint.ply", with_intensity=True)
    assert ctx.build_from_ply(0.01, tmp_path / "noint.ply").num_points == 4
    with pytest.raises(pcv.PcvError, match="Could not open"):
        ctx.build_from_ply(0.01, tmp_path / "missing.ply")
    with open(tmp_path / "short.ply", "wb") as f:  # the header promises more records than the file holds
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 400\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" + small.tobytes())
    with pytest.raises(pcv.PcvError, match="unexpected end of file"):
        ctx.build_from_ply(0.01, tmp_path / "short.ply")


def test_c_host_binary_builds_the_same_directory(tmp_path):
    """examples/build_octree.c (gcc -std=c11; the reference's src/bin/build_octree.rs:41-53 over the C ABI): PLY ->
    directory in a separate non-Python process, compared with the oracle's literal build."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "build_octree")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    x, y, z, rgb, _, _ = synthetic.gaussian_clusters(150_000, seed=43, num_clusters=4, extent=60.0, sigma_range=(0.05, 3.0))
    inten = (np.arange(x.size) % 23).astype(np.float32)
    with open(tmp_path / "cloud.ply", "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {x.size}\nproperty double x\nproperty double y\n"
                 "property double z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float intensity\n"
                 "end_header\n").encode())
        rec = np.zeros(x.size, dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("i", "<f4")])
        rec["x"], rec["y"], rec["z"], rec["r"], rec["g"], rec["b"], rec["i"] = x, y, z, rgb[:, 0], rgb[:, 1], rgb[:, 2], inten
        f.write(rec.tobytes())
    p = subprocess.run([exe, str(tmp_path / "cloud.ply"), "--output-directory", str(tmp_path / "c_out"), "--resolution", "0.001"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert f"{x.size} points" in p.stdout
    bmin, bmax = O.aabb(x, y, z)
    O.build_literal_dir(tmp_path / "cpu", 0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    assert not O.compare_octrees(O.load_dir(tmp_path / "c_out"), O.load_dir(tmp_path / "cpu"))


def test_ten_million_points_default_capacity_vs_oracle(ctx):
    """The reference's own constants (capacity 100 000, 1 mm) on a cloud large enough to need them: 10 M points,
    every node byte against the closed-form oracle."""
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(10_000_000, seed=77, num_clusters=16, extent=400.0,
                                                           sigma_range=(0.5, 12.0))
    want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=min(64, O.num_procs()))
    t = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb)
    assert t.num_nodes > 300 and max(t.node(i).level for i in range(t.num_nodes)) >= 6
    assert_same(t.to_dict(), want)


def test_copy_node_and_write_nodes(ctx, tmp_path):
    """pcv_octree_copy_node (device blob -> host / device buffer) and pcv_octree_write_nodes + pcv_write_meta, the pieces
    the multi-GPU output is assembled from, on a single tree: together they must reproduce pcv_octree_write_dir."""
    import torch
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(80_000, seed=13, num_clusters=4, extent=30.0, sigma_range=(0.05, 2.0))
    inten = (np.arange(x.size) % 311).astype(np.float32)
    t = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=2000)
    nodes = t.to_dict()
    for i in range(0, t.num_nodes, 7):
        nd = t.node(i)
        name = pcv.node_name(nd.id_high, nd.id_low)
        for which, key in enumerate(("xyz", "rgb", "intensity")):
            want = nodes[name][key]
            host = np.zeros(len(want) + 5, dtype=np.uint8)
            t.copy_node_into(i, which, host)
            assert host[:len(want)].tobytes() == want and not host[len(want):].any()
            dev = torch.zeros(len(want), dtype=torch.uint8, device="cuda")
            t.copy_node_into(i, which, dev)  # queued on the context's stream
            t.synchronize()
            assert dev.cpu().numpy().tobytes() == want
    with pytest.raises(pcv.PcvError):
        t.copy_node_into(0, 0, np.zeros(1, dtype=np.uint8))  # too small
    # write_nodes(level >= 1) + the root by hand + write_meta == write_dir
    a, b = tmp_path / "whole", tmp_path / "pieces"
    t.write_dir(str(a))
    t.write_nodes(str(b), 1)
    for ext in ("xyz", "rgb", "intensity"):
        (b / f"r.{ext}").write_bytes(nodes["r"][ext])
    table = [(t.node(i).id_high, t.node(i).id_low, t.node(i).num_points, t.node(i).encoding) for i in range(t.num_nodes)]
    pcv.octree.write_meta(str(b), 0.001, bmin, bmax, table)
    assert sorted(p.name for p in a.iterdir()) == sorted(p.name for p in b.iterdir())
    for p in a.iterdir():
        assert p.read_bytes() == (b / p.name).read_bytes(), p.name


def test_golden_fixtures(ctx):
    """HIP build against the frozen fixtures of tests/golden/ (node table + SHA-256 of every node file)."""
    import importlib.util
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    golden = json.load(open(os.path.join(here, "golden", "build_golden.json")))
    for case in G.CASES:
        x, y, z, rgb, inten, bmin, bmax, res, cap = G.make_case(case)
        got = ctx.build(res, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=cap).to_dict()
        assert G.digest(got) == golden[case[0]]["nodes"], case[0]
