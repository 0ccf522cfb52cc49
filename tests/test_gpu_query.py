"""Parity of the HIP transform-and-cull path (Q1-Q5) against the CPU oracle: prepared shapes (corners, axes),
Relation per (shape, node), relative_size_on_screen, get_visible_nodes order, nodes_in_location, per-point keep
masks on raw and on encoded node data, Isometry3 point transform. All exact (f64, same operation order)."""
import math

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


@pytest.fixture(scope="module")
def scene(ctx):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(300_000, seed=2, num_clusters=6, extent=100.0,
                                                           sigma_range=(0.5, 6.0))
    inten = (np.arange(x.size) % 251).astype(np.float32)
    tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=2000)
    with O.max_points_per_node(2000):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    return dict(x=x, y=y, z=z, bmin=bmin, bmax=bmax, tree=tree, oracle=want, names=tree.node_names())


def random_frusta(rng, bmin, bmax, n):
    out = []
    for _ in range(n):
        eye = rng.uniform(bmin - 20, bmax + 20)
        q = rng.normal(size=4)
        q = q / math.sqrt(float(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]))
        persp = O.perspective3_new(1.0, 1.2, 0.1, 100.0)  # BASELINE config 4
        c, qi = O.frustum_new(eye, q, persp)
        out.append((c, qi))
    return out


def test_prepared_shapes_match_oracle(ctx):
    rng = np.random.default_rng(1)
    fr = random_frusta(rng, np.zeros(3), np.full(3, 100.0), 50)
    shapes = [("frustum", c) for c, _ in fr] + [("frustum2", c, q) for c, q in fr[:10]]
    ax = np.array([0.2, 0.5, -0.7])
    ax = ax / math.sqrt((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2])
    obbs = [([0.0, 0, 0], [0.0, 0, 0, 1.0], [1.0, 2, 3]),
            ([0.0, 0, 0], O.quat_from_axis_angle([0, 0, 1.0], math.pi / 4), [1.0, 2, 3]),
            ([4.0, -2, 9], O.quat_from_axis_angle(ax.tolist(), 0.123), [1.0, 2, 3])]
    shapes += [("obb", t, q, h) for t, q, h in obbs]
    shapes += [("aabb", [0.5, 1.0, -3.0], [1.5, 3.0, 3.0]), ("frustum", np.zeros(16))]
    prepared = ctx.shapes(shapes)
    for i, sh in enumerate(shapes):
        corners, axes, valid = prepared.get(i)
        kind = {"frustum": O.SHAPE_FRUSTUM, "frustum2": O.SHAPE_FRUSTUM2, "obb": O.SHAPE_OBB, "aabb": O.SHAPE_AABB}[sh[0]]
        params = np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in sh[1:]])
        want = O.cached_axes(kind, params)
        if want is None:
            assert not valid
            continue
        assert valid
        assert np.array_equal(corners, want[0]), (i, sh[0])
        assert axes.shape == want[1].shape and np.array_equal(axes, want[1]), (i, sh[0])
    assert [prepared.get(60 + k)[1].shape[0] for k in range(3)] == [3, 5, 15]  # obb.rs:100-141 axis counts


def test_cull_nodes_relations_and_sizes(ctx, scene):
    rng = np.random.default_rng(3)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 40)
    tree = scene["tree"]
    prepared = ctx.shapes([("frustum", c) for c, _ in fr] + [("obb", scene["bmin"] + 30, [0, 0, 0.3, math.sqrt(1 - 0.09)], [20, 10, 5]),
                                                               ("aabb", scene["bmin"] + 10, scene["bmin"] + 50)])
    rel, sizes = tree.cull_nodes(prepared, with_sizes=True)
    cubes = np.array([[*tree.node(i).cube_min, tree.node(i).cube_edge] for i in range(tree.num_nodes)])
    for f, (c, _) in enumerate(fr):
        want_rel, want_sz = O.cull_cubes(O.SHAPE_FRUSTUM, c, cubes, with_sizes=True)
        assert np.array_equal(rel[f], want_rel), f
        assert np.array_equal(sizes[f], want_sz, equal_nan=True), f
    obb_params = list(scene["bmin"] + 30) + [0, 0, 0.3, math.sqrt(1 - 0.09)] + [20, 10, 5]
    assert np.array_equal(rel[40], O.cull_cubes(O.SHAPE_OBB, obb_params, cubes))
    assert np.array_equal(rel[41], O.cull_cubes(O.SHAPE_AABB, list(scene["bmin"] + 10) + list(scene["bmin"] + 50), cubes))
    assert {0, 1, 2} <= set(np.unique(rel))  # In, Cross and Out all occur


def test_cull_nodes_sparse_lists_equal_the_dense_matrix(ctx, scene):
    """pcv_cull_nodes_sparse: per shape the nodes whose sat() is not Out (sat.rs:174-194), in node order, with their Relation
    and relative_size_on_screen (octree/mod.rs:119-139) — against the oracle directly, and against a capacity that truncates."""
    rng = np.random.default_rng(13)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 48)
    tree = scene["tree"]
    m = tree.num_nodes
    prepared = ctx.shapes([("frustum", c) for c, _ in fr] + [("frustum", np.zeros(16)), ("aabb", scene["bmin"] + 10, scene["bmin"] + 50)])
    counts, idx, rel, sizes = tree.cull_nodes_sparse(prepared, m)
    cubes = np.array([[*tree.node(i).cube_min, tree.node(i).cube_edge] for i in range(m)])
    seen = 0
    for f, (c, _) in enumerate(fr):
        want_rel, want_sz = O.cull_cubes(O.SHAPE_FRUSTUM, c, cubes, with_sizes=True)
        keep = np.nonzero(want_rel != 2)[0]
        assert counts[f] == keep.size, f
        assert np.array_equal(idx[f, :keep.size], keep) and np.array_equal(rel[f, :keep.size], want_rel[keep]), f
        assert np.array_equal(sizes[f, :keep.size], want_sz[keep], equal_nan=True), f
        seen += keep.size
    assert seen > 100 and counts[48] == 0  # the singular matrix: no shape, nothing listed (dense: all Out)
    want = O.cull_cubes(O.SHAPE_AABB, list(scene["bmin"] + 10) + list(scene["bmin"] + 50), cubes)
    assert counts[49] == (want != 2).sum() and np.array_equal(idx[49, :counts[49]], np.nonzero(want != 2)[0])
    cap = max(1, int(counts.max()) // 2)  # truncation: the counts stay, the lists are the prefixes
    c2, i2, r2, _ = tree.cull_nodes_sparse(prepared, cap, with_sizes=False)
    assert np.array_equal(c2, counts)
    for f in range(prepared.count):
        k = min(int(counts[f]), cap)
        assert np.array_equal(i2[f, :k], idx[f, :k]) and np.array_equal(r2[f, :k], rel[f, :k])


def test_cull_nodes_sparse_tree_walk_never_differs_from_the_flat_evaluation(ctx, scene):
    """Round 6: the lists come from a walk down the tree (octree_iterator.rs:30-43: children only under a parent that is not
    Out) that skips a subtree only under a node that is Out BY A MARGIN; everything else is redone flat. Shapes built to sit
    ON the margin: boxes whose faces lie exactly on, one ulp inside and one ulp outside faces of node cubes (an Out node without
    the margin: flat redo), boxes that swallow the whole tree (more kept nodes than the wave's queue: flat redo), a box far away
    (the root Out by a margin: nothing listed) — each list must equal the dense matrix's row (pcv_cull_nodes), which evaluates
    every pair on its own."""
    tree = scene["tree"]
    m = tree.num_nodes
    nodes = [tree.node(i) for i in range(m)]
    # what the walk relies on: a node's cube in the table IS the recurrence step from its parent's cube in the table
    # (NodeId::find_bounding_cube, node.rs:160-170: edge /= 2; min += bit * edge) — bit for bit, up to the sign of a zero
    by_name = dict(zip(scene["names"], nodes))
    for name, nd in by_name.items():
        if name == "r":
            continue
        par, digit = by_name[name[:-1]], int(name[-1])
        half = par.cube_edge / 2.0
        want = [par.cube_min[0] + (half if digit & 4 else 0.0), par.cube_min[1] + (half if digit & 2 else 0.0),
                par.cube_min[2] + (half if digit & 1 else 0.0)]
        assert nd.cube_edge == half and list(nd.cube_min) == want, name
    deep = [nd for nd in nodes if nd.level >= 3][:: max(1, m // 40)][:40]
    shapes = []
    for nd in deep:
        lo = np.array(nd.cube_min)
        hi = lo + nd.cube_edge
        for step in (0, -1, 1):  # the box's max face ON / one ulp below / one ulp above the node's min face along x
            face = lo[0] if step == 0 else np.nextafter(lo[0], -np.inf if step < 0 else np.inf)
            shapes.append(("aabb", [face - 0.75 * nd.cube_edge, lo[1], lo[2]], [face, hi[1], hi[2]]))
        shapes.append(("aabb", hi, hi + 0.5 * nd.cube_edge))  # touching the node's max corner
    shapes.append(("aabb", scene["bmin"] - 1.0, scene["bmax"] + 1.0))          # everything In
    shapes.append(("aabb", scene["bmin"] + 1e-3, scene["bmax"] - 1e-3))        # nearly everything
    shapes.append(("aabb", scene["bmax"] + 1000.0, scene["bmax"] + 2000.0))    # nothing
    prepared = ctx.shapes(shapes)
    dense = tree.cull_nodes(prepared)
    counts, idx, rel, _ = tree.cull_nodes_sparse(prepared, m, with_sizes=False)
    redone = 0
    for f in range(len(shapes)):
        keep = np.nonzero(dense[f] != 2)[0]
        assert counts[f] == keep.size, (f, shapes[f])
        assert np.array_equal(idx[f, :keep.size], keep) and np.array_equal(rel[f, :keep.size], dense[f][keep]), (f, shapes[f])
        redone += keep.size > 1024
    assert counts[-1] == 0 and counts[-3] == m
    # a tree wide enough to overflow the wave's queue of 1 024 kept nodes: the whole-scene boxes are redone flat
    x, y, z = scene["x"], scene["y"], scene["z"]
    rgb = np.zeros((x.size, 3), np.uint8)
    big = ctx.build(0.001, pcv.Aabb(scene["bmin"], scene["bmax"]), x, y, z, rgb, max_points_per_node=120)
    mb = big.num_nodes
    assert mb > 4000
    dense = big.cull_nodes(prepared)
    counts, idx, rel, _ = big.cull_nodes_sparse(prepared, mb, with_sizes=False)
    for f in range(len(shapes)):
        keep = np.nonzero(dense[f] != 2)[0]
        assert counts[f] == keep.size, (f, shapes[f])
        assert np.array_equal(idx[f, :keep.size], keep) and np.array_equal(rel[f, :keep.size], dense[f][keep]), (f, shapes[f])
        redone += keep.size > 1024
    assert counts[-3] == mb and redone >= 2
    big.free()


def test_visible_nodes_match_reference_traversal_order(ctx, scene):
    rng = np.random.default_rng(4)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 64)
    tree, names = scene["tree"], scene["names"]
    vis, status = tree.visible_nodes(ctx.shapes([("frustum", c) for c, _ in fr] + [("frustum", np.zeros(16))]))
    nonempty = 0
    for f, (c, _) in enumerate(fr):
        want = O.get_visible_nodes(scene["bmin"], scene["bmax"], scene["oracle"].nodes, c)
        if want is None:
            assert status[f] != 0
            continue
        assert status[f] == 0
        assert [names[i] for i in vis[f]] == want, f  # same nodes, same BinaryHeap pop order
        nonempty += len(want) > 0
    assert nonempty > 10
    assert status[64] == 1 and len(vis[64]) == 0  # singular matrix: the reference panics


def test_nodes_in_location(ctx, scene):
    rng = np.random.default_rng(5)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 8)
    tree, names = scene["tree"], scene["names"]
    obb = (scene["bmin"] + 40, O.quat_from_axis_angle([0.0, 0.0, 1.0], 0.4), [25.0, 12.0, 30.0])
    shapes = [("all",), ("aabb", scene["bmin"] + 5, scene["bmin"] + 60), ("obb", *obb)] + [("frustum2", c, q) for c, q in fr]
    got = tree.nodes_in_location(ctx.shapes(shapes))
    on = scene["oracle"].nodes
    want = [O.nodes_in_location(scene["bmin"], scene["bmax"], on, O.SHAPE_ALL, None),
            O.nodes_in_location(scene["bmin"], scene["bmax"], on, O.SHAPE_AABB, list(scene["bmin"] + 5) + list(scene["bmin"] + 60)),
            O.nodes_in_location(scene["bmin"], scene["bmax"], on, O.SHAPE_OBB, list(obb[0]) + list(obb[1]) + list(obb[2]))]
    want += [O.nodes_in_location(scene["bmin"], scene["bmax"], on, O.SHAPE_FRUSTUM2, np.concatenate([c, q])) for c, q in fr]
    for g, w in zip(got, want):
        assert [names[i] for i in g] == w
    assert len(want[0]) == tree.num_nodes


def test_cull_points_raw_and_encoded(ctx, scene):
    rng = np.random.default_rng(6)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 4)
    obb = (scene["bmin"] + 40, O.quat_from_axis_angle([0.0, 1.0, 0.0], 0.9), [25.0, 12.0, 30.0])
    shapes = [("frustum", fr[0][0]), ("frustum2", *fr[1]), ("obb", *obb), ("aabb", scene["bmin"] + 20, scene["bmin"] + 70), ("all",)]
    kinds = [(O.SHAPE_FRUSTUM, fr[0][0]), (O.SHAPE_FRUSTUM, fr[1][0]), (O.SHAPE_OBB, list(obb[0]) + list(obb[1]) + list(obb[2])),
             (O.SHAPE_AABB, list(scene["bmin"] + 20) + list(scene["bmin"] + 70)), (O.SHAPE_ALL, None)]
    prepared = ctx.shapes(shapes)
    x, y, z = scene["x"], scene["y"], scene["z"]
    inten = (np.arange(x.size) % 251).astype(np.float32)
    total = 0
    for i, (kind, params) in enumerate(kinds):
        keep, kept = ctx.cull_points(prepared, i, x, y, z)
        want = O.cull_points(kind, params, x, y, z)
        assert np.array_equal(keep, want) and kept == int(want.sum()), i
        keep, kept = ctx.cull_points(prepared, i, x, y, z, inten, interval=(10.0, 99.5))
        want = O.cull_points(kind, params, x, y, z, inten, (10.0, 99.5))
        assert np.array_equal(keep, want) and kept == int(want.sum()), i
        total += kept
    assert total > 0
    # on a built octree's nodes: decode-on-the-fly == oracle decode + contains
    tree = scene["tree"]
    checked = 0
    for node in range(tree.num_nodes):
        nd = tree.node(node)
        if nd.num_points == 0:
            continue
        px, py, pz = O.decode_positions(nd.encoding, nd.cube_min, nd.cube_edge, tree.node_data(node, 0))
        for i, (kind, params) in enumerate(kinds[:4]):
            keep, kept = tree.cull_node_points(prepared, i, node)
            assert np.array_equal(keep, O.cull_points(kind, params, px, py, pz)), (node, i)
        ninten = np.frombuffer(tree.node_data(node, 2), dtype=np.float32)
        keep, kept = tree.cull_node_points(prepared, 2, node, interval=(0.0, 50.0))
        assert np.array_equal(keep, O.cull_points(kinds[2][0], kinds[2][1], px, py, pz, ninten, (0.0, 50.0)))
        checked += 1
        if checked >= 40:
            break
    assert checked >= 20


def test_transform_points(ctx, scene):
    iso = [3.5, -2.25, 10.0] + O.quat_from_axis_angle([0.6, 0.0, 0.8], 1.1)
    x, y, z = scene["x"][:100_000], scene["y"][:100_000], scene["z"][:100_000]
    ox, oy, oz = ctx.transform_points(iso, x, y, z)
    wx, wy, wz = O.iso_transform_points(iso, x, y, z)
    assert np.array_equal(ox, wx) and np.array_equal(oy, wy) and np.array_equal(oz, wz)


def test_open_dir_round_trip_and_query(ctx, scene, tmp_path):
    tree = scene["tree"]
    tree.write_dir(tmp_path / "oct")
    loaded = ctx.open_dir(tmp_path / "oct")
    assert loaded.num_nodes == tree.num_nodes and loaded.num_points == tree.num_points
    assert loaded.node_names() == scene["names"]
    m = loaded.meta()
    assert np.array_equal(m["bbox_min"], scene["bmin"]) and m["resolution"] == 0.001
    for i in (0, 1, tree.num_nodes // 2, tree.num_nodes - 1):
        a, b = tree.node(i), loaded.node(i)
        assert (a.num_points, a.encoding, tuple(a.cube_min), a.cube_edge) == (b.num_points, b.encoding, tuple(b.cube_min), b.cube_edge)
        assert loaded.node_data(i, 0) == tree.node_data(i, 0) and loaded.node_data(i, 1) == tree.node_data(i, 1)
    rng = np.random.default_rng(8)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 8)
    sh = ctx.shapes([("frustum", c) for c, _ in fr])
    va, _ = tree.visible_nodes(sh)
    vb, _ = loaded.visible_nodes(sh)
    assert all(np.array_equal(p, q) for p, q in zip(va, vb))
    with pytest.raises(pcv.PcvError):
        ctx.open_dir(tmp_path / "missing")


def check_query_points(scene, prepared, kinds, intervals=(None, (20.0, 180.0)), only=None):
    """pcv_query_points against: for node in nodes_in_location: decode, FilteredIterator keep mask, retain."""
    tree, on = scene["tree"], scene["oracle"].nodes
    index_of = {name: i for i, name in enumerate(scene["names"])}
    nonempty = 0
    for i, (kind, params) in enumerate(kinds):
        if only is not None and i not in only:
            continue
        for interval in intervals:
            got = tree.query_points(prepared, i, interval=interval)
            wx, wy, wz, wrgb, wint = [], [], [], [], []
            for name in O.nodes_in_location(scene["bmin"], scene["bmax"], on, kind, params):
                nd = on[name]
                if nd["num_points"] == 0:
                    continue
                info = tree.node(index_of[name])
                px, py, pz = O.decode_positions(nd["encoding"], info.cube_min, info.cube_edge, nd["xyz"])
                inten = np.frombuffer(nd["intensity"], dtype=np.float32)
                keep = O.cull_points(kind, params, px, py, pz, inten if interval else None, interval).astype(bool)
                wx.append(px[keep]); wy.append(py[keep]); wz.append(pz[keep])
                wrgb.append(np.frombuffer(nd["rgb"], dtype=np.uint8).reshape(-1, 3)[keep])
                wint.append(inten[keep])
            cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
            assert got["count"] == sum(len(p) for p in wx), (i, interval)
            assert np.array_equal(got["x"], cat(wx, np.float64)) and np.array_equal(got["y"], cat(wy, np.float64))
            assert np.array_equal(got["z"], cat(wz, np.float64))
            assert np.array_equal(got["rgb"].reshape(-1, 3), cat(wrgb, np.uint8).reshape(-1, 3))
            assert np.array_equal(got["intensity"], cat(wint, np.float32))
            nonempty += got["count"] > 0
    return nonempty


def test_query_points_batched(ctx, scene):
    """pcv_query_points == for node in nodes_in_location: decode, FilteredIterator keep mask, retain (iterator.rs)."""
    rng = np.random.default_rng(9)
    fr = random_frusta(rng, scene["bmin"], scene["bmax"], 6)
    obb = (scene["bmin"] + 45, O.quat_from_axis_angle([1.0, 0.0, 0.0], 0.5), [30.0, 20.0, 15.0])
    shapes = [("frustum2", *fr[i]) for i in range(6)] + [("obb", *obb), ("aabb", scene["bmin"] + 10, scene["bmin"] + 60), ("all",)]
    kinds = [(O.SHAPE_FRUSTUM2, np.concatenate(fr[i])) for i in range(6)]
    kinds += [(O.SHAPE_OBB, list(obb[0]) + list(obb[1]) + list(obb[2])),
              (O.SHAPE_AABB, list(scene["bmin"] + 10) + list(scene["bmin"] + 60)), (O.SHAPE_ALL, None)]
    prepared = ctx.shapes(shapes)
    tree = scene["tree"]
    assert check_query_points(scene, prepared, kinds) >= 6
    everything = tree.query_points(prepared, 8)
    assert everything["count"] == tree.num_points  # AllPoints returns the whole cloud
    small = tree.query_points(prepared, 8, capacity=1000)  # capacity smaller than the result
    assert small["count"] == tree.num_points and len(small["x"]) == 1000 and np.array_equal(small["x"], everything["x"][:1000])


def test_query_points_large(ctx):
    """The same parity on a cloud big enough for full-size chunks written by the device pre-pass (the small scene takes
    quarter-size chunks with host-built descriptors), with u8 / u16 / f32 nodes in one query and many partial chunks."""
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(5_000_000, seed=4, num_clusters=5, extent=400.0,
                                                           sigma_range=(1.0, 30.0))
    inten = (np.arange(x.size) % 251).astype(np.float32)
    tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=20000)
    with O.max_points_per_node(20000):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=8)
    scene = dict(bmin=bmin, bmax=bmax, tree=tree, oracle=want, names=tree.node_names())
    encodings = {tree.node(i).encoding for i in range(tree.num_nodes)}
    assert len(encodings) >= 2
    rng = np.random.default_rng(10)
    fr = random_frusta(rng, bmin, bmax, 1)
    lo, hi = bmin + (bmax - bmin) * 0.05, bmin + (bmax - bmin) * 0.9
    obb = ((bmin + bmax) / 2, O.quat_from_axis_angle([0.0, 0.0, 1.0], 0.3), list((bmax - bmin) * 0.3))
    shapes = [("all",), ("aabb", lo, hi), ("obb", *obb), ("frustum2", *fr[0])]
    kinds = [(O.SHAPE_ALL, None), (O.SHAPE_AABB, list(lo) + list(hi)),
             (O.SHAPE_OBB, list(obb[0]) + list(obb[1]) + list(obb[2])), (O.SHAPE_FRUSTUM2, np.concatenate(fr[0]))]
    prepared = ctx.shapes(shapes)
    assert check_query_points(scene, prepared, kinds, intervals=(None,), only=(0,)) == 1  # 5 M points: device descriptors
    assert check_query_points(scene, prepared, kinds, intervals=((20.0, 180.0),), only=(1, 2, 3)) >= 2
    assert tree.query_points(prepared, 0, capacity=1)["count"] == tree.num_points


def test_box_faces_on_decoded_positions_keep_the_reference_ties(ctx):
    """`mins <= p < maxs` (aabb.rs:46-48) on decoded positions: boxes whose faces are EXACTLY decoded point positions (p == min is
    inside, p == max is not), one ulp beside them, boxes that miss the cloud, an inverted and a NaN box — against the oracle's
    decode-and-compare, with and without the intensity interval. 3 M points: descriptors written on the device."""
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(3_000_000, seed=6, num_clusters=4, extent=300.0, sigma_range=(0.5, 20.0))
    inten = (np.arange(x.size) % 251).astype(np.float32)
    tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=20000)
    with O.max_points_per_node(20000):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=8)
    scene = dict(bmin=bmin, bmax=bmax, tree=tree, oracle=want, names=tree.node_names())
    # decoded positions of a few stored points (u16- and u8-coded nodes) as box faces
    index_of = {name: i for i, name in enumerate(scene["names"])}
    rng = np.random.default_rng(12)
    faces = []
    for name, nd in want.nodes.items():
        if nd["num_points"] > 100 and len(faces) < 6 and rng.random() < 0.02:
            info = tree.node(index_of[name])
            px, py, pz = O.decode_positions(nd["encoding"], info.cube_min, info.cube_edge, nd["xyz"])
            k = int(rng.integers(0, px.size))
            faces.append(np.array([px[k], py[k], pz[k]]))
    assert len(faces) >= 4
    boxes = []
    for a, b in zip(faces[0::2], faces[1::2]):
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        boxes += [(lo, hi), (np.nextafter(lo, np.inf), np.nextafter(hi, np.inf)), (np.nextafter(lo, -np.inf), np.nextafter(hi, -np.inf)),
                  (lo - 7.5, hi + 3.25)]
    boxes += [(bmax + 1.0, bmax + 2.0), (bmin + 50.0, bmin + 20.0), (np.array([np.nan, bmin[1], bmin[2]]), bmax),
              (bmin, np.array([bmax[0], np.nan, bmax[2]])), (bmin - 1.0, bmax + 1.0)]
    shapes = [("aabb", lo, hi) for lo, hi in boxes]
    kinds = [(O.SHAPE_AABB, list(lo) + list(hi)) for lo, hi in boxes]
    prepared = ctx.shapes(shapes)
    assert check_query_points(scene, prepared, kinds, intervals=(None, (20.0, 180.0))) >= 8
    assert tree.query_points(prepared, len(boxes) - 1, capacity=1)["count"] == tree.num_points
    tree.free()


def test_nodes_blob_matches_web_viewer_wire_format(ctx, scene):
    # octree_web_viewer/src/backend.rs:90-177
    import struct
    tree = scene["tree"]
    picks = [0, 1, tree.num_nodes // 3, tree.num_nodes - 1]
    want = b""

    def pad(b):
        return b + b"\0" * ((8 - len(b) % 8) % 8)

    for i in picks:
        nd = tree.node(i)
        bpc = {1: 1, 2: 2, 3: 4, 4: 8}[nd.encoding]
        head = struct.pack("<4dIB", *nd.cube_min, nd.cube_edge, nd.num_points, bpc)
        want = pad(want + head)
        want = pad(want + tree.node_data(i, 0))
        want = pad(want + tree.node_data(i, 1))
    assert tree.nodes_blob(picks) == want


def test_query_points_all_four_encodings(ctx):
    """City-scale extent at 1 mm: Float64 root / level 1, Float32, UInt16 and UInt8 levels below (codec.rs:31-40), a small
    capacity so that every encoding holds points; every chunk size of the staged decode (256 / 512 / 1 024 / 2 048
    points, quartered for this small query) against the oracle's decode + contains + retain."""
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(340_000, seed=12, num_clusters=6, extent=30000.0,
                                                           sigma_range=(5.0, 400.0), offset=(-2.7e6, -4.3e6, 3.8e6))
    rng = np.random.default_rng(13)  # one tight cluster: deep UInt8 levels
    c = np.array([x[0], y[0], z[0]])
    x = np.concatenate([x, c[0] + rng.normal(0.0, 0.03, 60_000)])
    y = np.concatenate([y, c[1] + rng.normal(0.0, 0.03, 60_000)])
    z = np.concatenate([z, c[2] + rng.normal(0.0, 0.03, 60_000)])
    rgb = synthetic.index_colors(x.size)
    bmin, bmax = np.array([x.min(), y.min(), z.min()]), np.array([x.max(), y.max(), z.max()])
    inten = (np.arange(x.size) % 251).astype(np.float32)
    tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=1500)
    with O.max_points_per_node(1500):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=8)
    scene = dict(bmin=bmin, bmax=bmax, tree=tree, oracle=want, names=tree.node_names())
    encodings = {tree.node(i).encoding for i in range(tree.num_nodes) if tree.node(i).num_points > 0}
    assert encodings == {1, 2, 3, 4}, encodings
    lo, hi = bmin + (bmax - bmin) * 0.1, bmin + (bmax - bmin) * 0.8
    shapes = [("all",), ("aabb", lo, hi)]
    kinds = [(O.SHAPE_ALL, None), (O.SHAPE_AABB, list(lo) + list(hi))]
    prepared = ctx.shapes(shapes)
    assert check_query_points(scene, prepared, kinds) >= 3
