"""Pin the query-path oracle against the reference's own unit tests."""
import math

import numpy as np

import oracle_lib as O
from point_cloud_viewer_amd import synthetic


def test_sat_cube_with_cube():
    # src/math/sat.rs:214-268
    c1 = [-1, -1, -1, 1, 1, 1]
    c2 = [-0.5, -0.5, -0.5, 1.5, 1.5, 1.5]
    c3 = [-0.9, -0.9, -0.9, -0.7, -0.7, -0.7]
    A = O.SHAPE_AABB
    assert O.intersect_shapes(A, c1, A, c2) == O.REL_CROSS
    assert O.intersect_shapes(A, c2, A, c3) == O.REL_OUT
    assert O.intersect_shapes(A, c1, A, c3) == O.REL_IN
    assert O.intersect_shapes(A, c3, A, c1) == O.REL_CROSS


def test_obb_intersects_aabb():
    # src/geometry/obb.rs:100-141: cached-axis counts 3 / 5 / 15, relations Cross / Out
    half = [1.0, 2.0, 3.0]
    bbox = [0.5, 1.0, -3.0, 1.5, 3.0, 3.0]
    cube = np.array([[0, 0, 0, 0.0]])  # unused placeholder

    def obb(q):
        return [0.0, 0.0, 0.0] + list(q) + half

    def relation(params):
        corners, axes = O.cached_axes(O.SHAPE_OBB, params)
        bc = np.array([[bbox[0], bbox[1], bbox[2]], [bbox[3], bbox[1], bbox[2]], [bbox[0], bbox[4], bbox[2]],
                       [bbox[3], bbox[4], bbox[2]], [bbox[0], bbox[1], bbox[5]], [bbox[3], bbox[1], bbox[5]],
                       [bbox[0], bbox[4], bbox[5]], [bbox[3], bbox[4], bbox[5]]], dtype=np.float64)
        rel = O.lib().pcvo_sat_raw(len(axes), O._d(np.ascontiguousarray(axes)), O._d(np.ascontiguousarray(corners)), 8,
                                   O._d(np.ascontiguousarray(bc)), 8)
        return len(axes), rel

    n, rel = relation(obb([0.0, 0.0, 0.0, 1.0]))
    assert n == 3 and rel == O.REL_CROSS
    n, rel = relation(obb(O.quat_from_axis_angle([0.0, 0.0, 1.0], math.pi / 4.0)))
    assert n == 5 and rel == O.REL_OUT
    ax = np.array([0.2, 0.5, -0.7])
    ax = ax / math.sqrt((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2])
    n, _ = relation(obb(O.quat_from_axis_angle(ax.tolist(), 0.123)))
    assert n == 15


def test_perspective_matches_nalgebra_perspective3():
    # src/geometry/frustum.rs:178-205 compare_perspective (exact element equality)
    aspect, fovy, near, far = 1.2, 0.66, 1.0, 100.0
    ymax = near * math.tan(fovy * 0.5)
    xmax = ymax * aspect
    a = O.perspective_new(-xmax, xmax, -ymax, ymax, near, far)
    b = O.perspective3_new(aspect, fovy, near, far)
    assert np.array_equal(a, b)


def test_perspective_inverse():
    # src/math/mod.rs:191-198
    p = O.perspective_new(-0.123, 0.45, 0.04, 0.75, 1.0, 4.0)
    ref = O.mat4_try_inverse(p)
    assert ref is not None
    assert np.abs(ref - O.perspective_inverse(p)).max() < 1e-6
    assert O.mat4_try_inverse(np.zeros(16)) is None


def test_frustum_intersects_aabb():
    # src/math/mod.rs:200-220
    rot = O.quat_from_axis_angle([1.0, 0.0, 0.0], math.pi)
    persp = O.perspective_new(-0.5, 0.0, -0.5, 0.0, 1.0, 4.0)
    c, q = O.frustum_new([0, 0, 0], rot, persp)
    bmin, bmax = [-0.5, 0.25, 1.5], [-0.25, 0.5, 3.5]
    assert O.intersect_shapes(O.SHAPE_FRUSTUM2, np.concatenate([c, q]), O.SHAPE_AABB, bmin + bmax) == O.REL_IN
    keep = O.cull_points(O.SHAPE_FRUSTUM, c, [bmin[0], bmax[0]], [bmin[1], bmax[1]], [bmin[2], bmax[2]])
    assert keep.tolist() == [1, 1]
    # analytic inverse and cofactor inverse describe the same frustum
    assert np.abs(O.mat4_try_inverse(c) - q).max() < 1e-9


def test_point_culling_equals_sat_on_a_point():
    # point_cloud_test/tests/main.rs:104-127: contains(p) == (sat(face_normals, corners, [p]) == In)
    rng = np.random.default_rng(5)
    rot = O.quat_from_axis_angle([0.0, 0.0, 1.0], 0.7)
    c, q = O.frustum_new([5.0, -3.0, 2.0], rot, O.perspective3_new(1.0, 1.2, 0.1, 100.0))
    p = rng.uniform(-60, 60, (20000, 3))
    keep = O.cull_points(O.SHAPE_FRUSTUM, c, p[:, 0], p[:, 1], p[:, 2])
    # face normals = first 5 cached axes are the frustum's own normals (separating_axes_iter order)
    corners, axes = O.cached_axes(O.SHAPE_FRUSTUM2, np.concatenate([c, q]))
    normals = np.ascontiguousarray(axes[:5])
    mismatch = 0
    for i in range(0, 20000, 7):
        rel = O.lib().pcvo_sat_raw(5, O._d(normals), O._d(np.ascontiguousarray(corners)), 8,
                                   O._d(np.ascontiguousarray(p[i])), 1)
        mismatch += int((rel == O.REL_IN) != bool(keep[i]))
    assert mismatch == 0
    assert 0 < keep.sum() < keep.size


def test_visible_nodes_and_location_queries_are_consistent():
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(200_000, seed=2, num_clusters=6, extent=100.0,
                                                           sigma_range=(0.5, 6.0))
    with O.max_points_per_node(2000):
        t = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    assert len(t.nodes) > 100
    everything = O.nodes_in_location(bmin, bmax, t.nodes, O.SHAPE_ALL, None)
    assert sorted(everything) == sorted(t.nodes)
    assert everything[0] == "r" and [len(n) for n in everything] == sorted(len(n) for n in everything)  # BFS order
    # a frustum looking at the cloud from outside sees a strict, non-empty subset; every ancestor of a visible
    # node is visible or empty (hierarchical traversal, octree/mod.rs:228-283)
    eye = [bmin[0] - 30.0, (bmin[1] + bmax[1]) / 2, (bmin[2] + bmax[2]) / 2]
    rot = O.quat_from_axis_angle([0.0, 1.0, 0.0], -math.pi / 2)  # look along +x
    c, q = O.frustum_new(eye, rot, O.perspective3_new(1.0, 0.6, 0.1, 400.0))
    vis = O.get_visible_nodes(bmin, bmax, t.nodes, c)
    assert vis is not None and 0 < len(vis) < len(t.nodes)
    assert len(set(vis)) == len(vis)
    vs = set(vis)
    for n in vis:
        if len(n) > 1:
            assert n[:-1] in vs or t.nodes[n[:-1]]["num_points"] == 0
    assert all(t.nodes[n]["num_points"] > 0 for n in vis)
    # the same frustum as a PointLocation: nodes_in_location tests every node's own cube (no In short-cut, no
    # emptiness filter) -> superset of the visible set
    loc = O.nodes_in_location(bmin, bmax, t.nodes, O.SHAPE_FRUSTUM2, np.concatenate([c, q]))
    assert vs <= set(loc)
    inside = O.nodes_in_location(bmin, bmax, t.nodes, O.SHAPE_AABB, list(bmin - 1) + list(bmax + 1))
    assert sorted(inside) == sorted(t.nodes)
