"""Seeded random shapes against the CPU oracle on the node cubes of a built octree: hundreds of oriented boxes (random
unit quaternions, half extents from a millimetre to larger than the cloud, one of them exactly zero), axis-aligned boxes
whose faces lie ON node cube planes (the octree's cubes are min + k * edge / 2^L: `contains` / SAT ties, aabb.rs:46-48,
sat.rs:55-101) or one ulp beside them, empty and inverted boxes, and frusta with narrow / wide fields of view and near
planes from 1e-3 to 10. Relation for Relation like the fixed-seed tests of test_gpu_query.py, and the per-point keep
masks of a few of each kind."""
import math

import numpy as np
import pytest

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import synthetic
from test_gpu_query import check_query_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pcv.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def scene(ctx):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(200_000, seed=12, num_clusters=5, extent=64.0, sigma_range=(0.3, 5.0))
    # a power-of-two cube at the origin: every node cube plane is an exactly representable number
    bmin, bmax = np.zeros(3), np.full(3, 64.0)
    x, y, z = (np.clip(v, 0.0, 64.0) for v in (x, y, z))
    inten = (np.arange(x.size) % 251).astype(np.float32)
    tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=1500)
    with O.max_points_per_node(1500):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    cubes = np.array([[*tree.node(i).cube_min, tree.node(i).cube_edge] for i in range(tree.num_nodes)])
    return dict(tree=tree, cubes=cubes, bmin=bmin, bmax=bmax, oracle=want, names=tree.node_names())


def _unit_quat(rng):
    q = rng.normal(size=4)
    return q / math.sqrt(float(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]))


def _plane(rng):
    """A coordinate on a node cube plane of a random level, sometimes an ulp beside it."""
    level = int(rng.integers(0, 9))
    v = float(rng.integers(0, 2 ** level + 1)) * (64.0 / 2 ** level)
    step = int(rng.integers(-1, 2))
    return float(np.nextafter(v, math.inf if step > 0 else -math.inf)) if step else v


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_relations_equal_oracle(ctx, scene, seed):
    rng = np.random.default_rng(500 + seed)
    shapes, want_kind = [], []
    for _ in range(120):  # oriented boxes
        centre = rng.uniform(-10.0, 74.0, 3)
        half = 10.0 ** rng.uniform(-3, 2, 3)
        if rng.random() < 0.1:
            half[int(rng.integers(0, 3))] = 0.0
        q = _unit_quat(rng) if rng.random() < 0.9 else np.array([0.0, 0.0, 0.0, 1.0])
        shapes.append(("obb", centre, q, half))
        want_kind.append((O.SHAPE_OBB, np.concatenate([centre, q, half])))
    for _ in range(120):  # axis-aligned boxes on / beside the cube planes
        lo = np.array([_plane(rng) for _ in range(3)])
        hi = np.array([_plane(rng) for _ in range(3)])
        if rng.random() < 0.85:
            lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)  # the rest stay inverted / empty
        shapes.append(("aabb", lo, hi))
        want_kind.append((O.SHAPE_AABB, np.concatenate([lo, hi])))
    for _ in range(60):  # frusta
        eye = rng.uniform(-30.0, 94.0, 3)
        persp = O.perspective3_new(float(rng.choice([0.5, 1.0, 1.7777])), float(rng.uniform(0.05, 2.8)),
                                   float(10.0 ** rng.uniform(-3, 1)), float(10.0 ** rng.uniform(1.1, 3)))
        c, _ = O.frustum_new(eye, _unit_quat(rng), persp)
        shapes.append(("frustum", c))
        want_kind.append((O.SHAPE_FRUSTUM, np.asarray(c, dtype=np.float64).ravel()))
    prepared = ctx.shapes(shapes)
    rel = scene["tree"].cull_nodes(prepared)
    # round 6: the per-shape lists come from a walk down the tree that may skip a subtree only under a node that is Out by a
    # margin — boxes ON and one ulp beside the cube planes are exactly where that margin decides; the lists must be the rows'
    counts, idx, srel, _ = scene["tree"].cull_nodes_sparse(prepared, scene["tree"].num_nodes, with_sizes=False)
    seen = set()
    for i, (kind, params) in enumerate(want_kind):
        want = O.cull_cubes(kind, params, scene["cubes"])
        assert np.array_equal(rel[i], want), (seed, i, shapes[i][0])
        keep = np.nonzero(want != 2)[0]
        assert counts[i] == keep.size and np.array_equal(idx[i, :keep.size], keep) and np.array_equal(srel[i, :keep.size], want[keep]), \
            (seed, i, shapes[i][0], "sparse list")
        seen |= set(np.unique(want).tolist())
    assert {0, 1, 2} <= seen
    # ... and PointCloud::nodes_in_location (octree/mod.rs:309-323) of every shape — the same walk, pruning under EVERY Out node —
    # against the oracle's NodeIdsIterator
    got = scene["tree"].nodes_in_location(prepared)
    on, names = scene["oracle"].nodes, scene["names"]
    for i, (kind, params) in enumerate(want_kind):
        assert [names[k] for k in got[i]] == O.nodes_in_location(scene["bmin"], scene["bmax"], on, kind, params), (seed, i, shapes[i][0])


@pytest.mark.parametrize("seed", range(3))
def test_random_shapes_points_equal_oracle(ctx, scene, seed):
    """pcv_query_points of random shapes == nodes_in_location + decode + keep mask + retain (iterator.rs:96-119, 226-333):
    the points of the clipped cloud lie ON the root cube's faces by the thousand, the boxes' faces on node cube planes."""
    rng = np.random.default_rng(900 + seed)
    shapes, kinds = [], []
    for _ in range(6):
        centre, half, q = rng.uniform(0.0, 64.0, 3), 10.0 ** rng.uniform(-0.5, 1.6, 3), _unit_quat(rng)
        shapes.append(("obb", centre, q, half))
        kinds.append((O.SHAPE_OBB, list(centre) + list(q) + list(half)))
    for _ in range(6):
        lo = np.array([_plane(rng) for _ in range(3)])
        hi = np.array([_plane(rng) for _ in range(3)])
        lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
        shapes.append(("aabb", lo, hi))
        kinds.append((O.SHAPE_AABB, list(lo) + list(hi)))
    for _ in range(4):
        persp = O.perspective3_new(1.0, float(rng.uniform(0.3, 2.0)), float(10.0 ** rng.uniform(-2, 0)), float(10.0 ** rng.uniform(1.3, 2.5)))
        c, qi = O.frustum_new(rng.uniform(-10.0, 74.0, 3), _unit_quat(rng), persp)
        shapes.append(("frustum2", c, qi))
        kinds.append((O.SHAPE_FRUSTUM2, np.concatenate([np.asarray(c, dtype=np.float64).ravel(), np.asarray(qi, dtype=np.float64).ravel()])))
    assert check_query_points(scene, ctx.shapes(shapes), kinds) >= 6
