"""Seeded random builds against the CPU oracle, aimed at the places where the reference's arithmetic is decided by a tie:
points ON the octant planes of many levels (`p > centre` is strict, node.rs:34-42), points one or two ulps beside them,
codes that are exact integers before the truncating cast (codec.rs:102-113), few distinct positions repeated many times,
flat and one-dimensional clouds, cubes far from the origin, loose and tight bounding boxes, capacities from a dozen points
to thousands — through the exact pipeline and through the forced single-chain build. Byte-exact like every other build
test (src/octree/generation.rs:289-403 has one answer per input)."""
import numpy as np
import pytest

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from test_gpu_build import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pcv.Context(0)
    yield c
    c.close()


def _cloud(rng, kind, n, lo, edge):
    """n points in (mostly inside) the cube [lo, lo + edge)^3."""
    if kind == "uniform":
        p = lo + rng.random((n, 3)) * edge
    elif kind == "clusters":
        k = int(rng.integers(1, 9))
        centres = lo + rng.random((k, 3)) * edge
        sig = edge * 10.0 ** rng.uniform(-4, -1, k)
        which = rng.integers(0, k, n)
        p = centres[which] + rng.standard_normal((n, 3)) * sig[which, None]
        p = np.clip(p, lo, lo + edge)
    elif kind in ("lattice", "near_lattice"):
        # k * edge / 2^L: the planes the octants of level L are cut by (and, for the fix-point codes, positions whose
        # quotient times 255 / 65535 is close to an integer)
        levels = rng.integers(1, 14, (n, 3))
        cells = np.floor(rng.random((n, 3)) * (2.0 ** levels))
        p = lo + cells * (edge / 2.0 ** levels)
        if kind == "near_lattice":
            steps = rng.integers(-2, 3, (n, 3))
            for _ in range(2):
                up = steps > 0
                p = np.where(up, np.nextafter(p, np.inf), np.where(steps < 0, np.nextafter(p, -np.inf), p))
                steps = steps - np.sign(steps)
    elif kind == "duplicates":
        distinct = lo + rng.random((int(rng.integers(1, 40)), 3)) * edge
        p = distinct[rng.integers(0, distinct.shape[0], n)]
    elif kind == "plane":
        p = lo + rng.random((n, 3)) * edge
        p[:, int(rng.integers(0, 3))] = lo[0] + edge * float(rng.choice([0.0, 0.25, 0.5, 0.5000001, 0.999]))
    elif kind == "line":
        t = rng.random(n)
        p = lo + np.stack([t, t, t], axis=1) * edge
    else:
        raise AssertionError(kind)
    return np.ascontiguousarray(p[:, 0]), np.ascontiguousarray(p[:, 1]), np.ascontiguousarray(p[:, 2])


KINDS = ["uniform", "clusters", "lattice", "near_lattice", "duplicates", "plane", "line"]
CASES = [(seed, KINDS[seed % len(KINDS)]) for seed in range(140)]


@pytest.mark.parametrize("seed,kind", CASES)
def test_random_build_equals_oracle(ctx, seed, kind):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 9, 257, 5_000, 40_000, 150_000], p=[0.03, 0.03, 0.04, 0.1, 0.2, 0.3, 0.3]))
    origin = np.array(rng.choice([0.0, -3.5, 1.0e3, 4.2e6, -6.3e6], 3), dtype=np.float64) + rng.random(3)
    edge = float(rng.choice([2.0 ** -6, 1.0, 10.0, 37.3, 256.0, 1000.0, 65536.0]))
    x, y, z = _cloud(rng, kind, n, origin, edge)
    if rng.random() < 0.5:  # the cube the points were drawn in, possibly padded
        pad = edge * float(rng.choice([0.0, 0.0, 0.01, 0.5]))
        bmin, bmax = origin - pad, origin + edge + pad
    else:  # the tight box of the points (find_bounding_box, generation.rs:256-270)
        bmin = np.array([x.min(), y.min(), z.min()])
        bmax = np.array([x.max(), y.max(), z.max()])
    # resolutions between a 2^7 and a 2^19 fraction of the edge: u8 / u16 / Float32-coded levels in every mix; with
    # far-away origins also Float64-coded ones
    res = max(float(np.max(bmax - bmin)), edge * 1e-9) / 2.0 ** float(rng.uniform(7, 19))
    cap = int(rng.choice([12, 100, 1_000, 5_000]))
    rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    inten = rng.random(n).astype(np.float32) if rng.random() < 0.3 else None
    with O.max_points_per_node(cap):
        want = O.build_closed(res, bmin, bmax, x, y, z, rgb, inten, threads=4)
    for single_chain in (False, True):
        t = ctx.build(res, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=cap, single_chain=single_chain)
        assert_same(t.to_dict(), want, check_intensity=inten is not None)
        t.free()


# ---- the same generator at sizes where the record sort takes its two-pass form: the second pass writes the node bytes of the
# integer-coded leaves itself (downsweep_settle_kernel) — ties on octant planes, repeated positions and far-away origins through
# THAT rewrite (generation.rs:222-238), with and without the intensity plane
SETTLED = {"cases": 0, "settled": 0}
BIG_CASES = [(seed, KINDS[seed % len(KINDS)]) for seed in range(28)]


@pytest.mark.parametrize("seed,kind", BIG_CASES)
def test_random_build_through_the_settling_sort_pass(ctx, seed, kind):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([300_000, 600_000]))
    origin = np.array(rng.choice([0.0, -3.5, 1.0e3, 4.2e6, -6.3e6], 3), dtype=np.float64) + rng.random(3)
    edge = float(rng.choice([1.0, 37.3, 256.0, 1000.0]))
    x, y, z = _cloud(rng, kind, n, origin, edge)
    pad = edge * float(rng.choice([0.0, 0.01, 0.5]))
    bmin, bmax = origin - pad, origin + edge + pad
    res = edge / 2.0 ** float(rng.uniform(9, 18))
    cap = int(rng.choice([150, 400, 1_500]))
    rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    inten = rng.random(n).astype(np.float32) if rng.random() < 0.4 else None
    with O.max_points_per_node(cap):
        want = O.build_closed(res, bmin, bmax, x, y, z, rgb, inten, threads=8)
    t = ctx.build(res, pcv.Aabb(bmin, bmax), x, y, z, rgb, inten, max_points_per_node=cap, single_chain=True)
    info = t.build_info()
    assert_same(t.to_dict(), want, check_intensity=inten is not None)
    t.free()
    SETTLED["cases"] += 1
    SETTLED["settled"] += 1 if info["settled_in_sort"] > 0 else 0


def test_the_cases_above_went_through_the_settling_pass():
    """(duplicates / line / plane clouds have few leaves or fall back to the exact pipeline: not every case can)"""
    assert SETTLED["cases"] == len(BIG_CASES) and SETTLED["settled"] >= SETTLED["cases"] // 3, SETTLED
