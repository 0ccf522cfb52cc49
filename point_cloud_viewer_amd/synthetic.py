"""Seeded synthetic point clouds for the BASELINE.json configs (host side, numpy).

The reference's own generator (point_cloud_test/src/synthetic_data.rs:22-83) draws from rand 0.7.3
StdRng, which is not reproducible here; these generators keep its *distribution* and bbox rule and
use numpy's PCG64 instead.
"""
import math

import numpy as np

# WGS84 (nav-types 0.5.1 constants, used by reference src/math/mod.rs:167-183)
_WGS84_A = 6378137.0
_WGS84_F = 1.0 / 298.257223563


def _ecef_from_lat_lng(lat_deg, lon_deg):
    lat, lon = math.radians(lat_deg), math.radians(lon_deg)
    e2 = _WGS84_F * (2.0 - _WGS84_F)
    n = _WGS84_A / math.sqrt(1.0 - e2 * math.sin(lat) ** 2)
    return np.array([n * math.cos(lat) * math.cos(lon), n * math.cos(lat) * math.sin(lon),
                     n * (1.0 - e2) * math.sin(lat)])


def ecef_from_local(lat_deg, lon_deg):
    """(R, t) with p_ecef = R @ p_enu + t — the inverse of reference `local_frame_from_lat_lng`
    (src/math/mod.rs:167-183): ENU axes at the given geodetic origin."""
    lat, lon = math.radians(lat_deg), math.radians(lon_deg)
    east = np.array([-math.sin(lon), math.cos(lon), 0.0])
    north = np.array([-math.sin(lat) * math.cos(lon), -math.sin(lat) * math.sin(lon), math.cos(lat)])
    up = np.array([math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)])
    rot = np.stack([east, north, up], axis=1)
    return rot, _ecef_from_lat_lng(lat_deg, lon_deg)


def index_colors(n):
    """rgb = 24-bit point index (synthetic_data.rs:68-73)."""
    i = np.arange(n, dtype=np.uint32)
    return np.stack([(i >> 16) & 255, (i >> 8) & 255, i & 255], axis=1).astype(np.uint8)


def hash_colors(n, with_alpha=False):
    """rgb(a) = a cheap integer hash of the index."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
    h = h.astype(np.uint32)
    cols = [(h >> 16) & 255, (h >> 8) & 255, h & 255]
    if with_alpha:
        cols.append((h >> 24) & 255)
    return np.stack(cols, axis=1).astype(np.uint8)


def uniform_ecef(n, seed=80293751232, width=200.0, height=20.0, lat=37.407204, lon=-122.147604):
    """BASELINE config 1: uniform box in an ENU frame placed in ECEF; bbox = the transformed local box
    (loose, as synthetic_data.rs:46-50 via Aabb::transform aabb.rs:58-66)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    hw, hh = width * 0.5, height * 0.5
    local = np.empty((n, 3))
    local[:, 0] = rng.uniform(-hw, hw, n)
    local[:, 1] = rng.uniform(-hw, hw, n)
    local[:, 2] = rng.uniform(-hh, hh, n)
    rot, t = ecef_from_local(lat, lon)
    p = local @ rot.T + t
    corners = np.array([[sx * hw, sy * hw, sz * hh] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)])
    c = corners @ rot.T + t
    return (np.ascontiguousarray(p[:, 0]), np.ascontiguousarray(p[:, 1]), np.ascontiguousarray(p[:, 2]),
            index_colors(n), c.min(axis=0), c.max(axis=0))


def gaussian_clusters(n, seed=1, num_clusters=64, extent=1000.0, sigma_range=(1.0, 20.0), offset=(0.0, 0.0, 0.0),
                      chunk=1 << 22):
    """BASELINE config 2/3/5: equal-weight isotropic Gaussian clusters, centres uniform in a cube.
    Points are emitted in a cluster-interleaved (random) order. bbox = exact min/max."""
    rng = np.random.Generator(np.random.PCG64(seed))
    centres = rng.uniform(0.0, extent, (num_clusters, 3)) + np.asarray(offset, dtype=np.float64)
    sigmas = rng.uniform(sigma_range[0], sigma_range[1], num_clusters)
    x = np.empty(n)
    y = np.empty(n)
    z = np.empty(n)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        which = rng.integers(0, num_clusters, m)
        g = rng.standard_normal((m, 3)) * sigmas[which, None] + centres[which]
        x[s:s + m] = g[:, 0]
        y[s:s + m] = g[:, 1]
        z[s:s + m] = g[:, 2]
    bmin = np.array([x.min(), y.min(), z.min()]) if n else np.zeros(3)
    bmax = np.array([x.max(), y.max(), z.max()]) if n else np.zeros(3)
    return x, y, z, hash_colors(n), bmin, bmax


def reference_unit_test_cloud():
    """reference src/octree/tests.rs:18-46: 100 000 points at the origin + one at (-200,-40,30),
    colour (255,0,0), bbox = Aabb of those two points, resolution 1.0."""
    n = 100001
    x = np.zeros(n)
    y = np.zeros(n)
    z = np.zeros(n)
    x[-1], y[-1], z[-1] = -200.0, -40.0, 30.0
    rgb = np.zeros((n, 3), dtype=np.uint8)
    rgb[:, 0] = 255
    bmin = np.array([-200.0, -40.0, 0.0])
    bmax = np.array([0.0, 0.0, 30.0])
    return x, y, z, rgb, bmin, bmax, 1.0
