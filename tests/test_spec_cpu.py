"""Host logic of the single-chain build (point_cloud_viewer_amd/csrc/pcv_spec.cpp) on the CPU: sample tree -> predicted
tree T'' -> walk of every point -> exact counts -> true tree, driven by the ORACLE's full-depth path keys through the
test hook pcv_spec_selftest. The true tree must be the oracle's tree whenever the status is OK; a prediction that does
not cover the decision must be reported (the build then falls back to the exact pipeline), never papered over."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import synthetic


def _selftest(keys, stride, cap, delta, resolution, edges, nlevels, force_mask=0):
    lib = pcv.load_library()
    f = lib.pcv_spec_selftest
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_uint32,
                  C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    capn = 1 << 16
    prefix, level = np.zeros(capn, dtype=np.uint64), np.zeros(capn, dtype=np.uint8)
    count, opn = np.zeros(capn, dtype=np.uint64), np.zeros(capn, dtype=np.uint8)
    num, stats = C.c_uint64(0), np.zeros(4, dtype=np.uint64)
    e = np.ascontiguousarray(edges, dtype=np.float64)
    rc = f(keys.ctypes.data, keys.size, stride, cap, delta, resolution, e.ctypes.data, nlevels, force_mask, capn,
           prefix.ctypes.data, level.ctypes.data, count.ctypes.data, opn.ctypes.data, C.byref(num), stats.ctypes.data)
    m = num.value
    return rc, prefix[:m], level[:m], count[:m], opn[:m], stats


def _name(prefix, level):
    return "r" + "".join(str((int(prefix) >> (3 * (21 - k))) & 7) for k in range(1, int(level) + 1))


@pytest.mark.parametrize("n,cap,stride,delta,seed", [(300_000, 2000, 16, 0.25, 1), (300_000, 2000, 32, 0.4, 2),
                                                      (200_000, 500, 8, 0.3, 3), (400_000, 5000, 32, 0.3, 4),
                                                      (150_000, 1000, 1, 0.0, 5), (250_000, 100_000, 32, 0.3, 6)])
def test_true_tree_from_sampled_prediction_is_the_oracle_tree(n, cap, stride, delta, seed):
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=seed, num_clusters=5, extent=120.0, sigma_range=(0.05, 6.0))
    ml, edges, _ = O.level_table(bmin, bmax, 0.001)
    nlevels = min(ml, 21)
    keys = O.chain_keys64(bmin, bmax, 0.001, nlevels, x, y, z, threads=4)
    rc, prefix, level, count, opn, stats = _selftest(keys, stride, cap, delta, 0.001, edges, nlevels)
    assert rc == 0, rc
    with O.max_points_per_node(cap):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    got = {_name(p, l): (int(c), int(o)) for p, l, c, o in zip(prefix, level, count, opn)}
    assert set(got) == set(want.nodes)
    sk = np.sort(keys)
    for name, (cnt, is_open) in got.items():
        lvl = len(name) - 1
        pfx = sum(int(d) << (3 * (21 - k)) for k, d in enumerate(name[1:], start=1))
        lo = np.searchsorted(sk, np.uint64(pfx), "left")
        hi = np.searchsorted(sk, np.uint64(pfx + (1 << (3 * (21 - lvl))) - 1) if lvl else np.uint64((1 << 63) - 1), "right")
        assert cnt == hi - lo, name
        children = [k for k in want.nodes if len(k) == len(name) + 1 and k.startswith(name)]
        assert bool(is_open) == bool(children), name
    if stride > 1:
        assert stats[1] >= sum(1 for v in got.values() if not v[1])  # at least as many predicted leaves as true leaves


def test_a_prediction_that_is_too_shallow_is_reported():
    """No band at all and a coarse sample: some predicted leaf really holds more than the capacity -> not OK."""
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(300_000, seed=9, num_clusters=6, extent=100.0, sigma_range=(0.05, 5.0))
    ml, edges, _ = O.level_table(bmin, bmax, 0.001)
    nlevels = min(ml, 21)
    keys = O.chain_keys64(bmin, bmax, 0.001, nlevels, x, y, z, threads=4)
    seen = set()
    for stride in (64, 128, 256):
        rc, *_ = _selftest(keys, stride, 1500, 0.0, 0.001, edges, nlevels)
        seen.add(rc)
    assert 1 in seen and seen <= {0, 1}, seen  # PCV_SPEC_TOO_SHALLOW: the build falls back to the exact pipeline


def test_forced_level1_split_and_duplicates():
    rng = np.random.default_rng(3)
    n = 120_000
    x = np.concatenate([np.zeros(n // 2), rng.uniform(-50, 50, n // 2)])
    y = np.concatenate([np.zeros(n // 2), rng.uniform(-50, 50, n // 2)])
    z = np.concatenate([np.zeros(n // 2), rng.uniform(-5, 5, n // 2)])
    perm = rng.permutation(n)
    x, y, z = x[perm], y[perm], z[perm]
    bmin, bmax = np.array([-50.0, -50, -5]), np.array([50.0, 50, 5])
    ml, edges, _ = O.level_table(bmin, bmax, 0.5)
    nlevels = min(ml, 21)
    keys = O.chain_keys64(bmin, bmax, 0.5, nlevels, x, y, z, threads=2)
    rgb = np.zeros((n, 3), dtype=np.uint8)
    for force in (0, 0b10010001):
        rc, prefix, level, count, opn, _ = _selftest(keys, 16, 3000, 0.3, 0.5, edges, nlevels, force)
        assert rc == 0
        got = {_name(p, l) for p, l in zip(prefix, level)}
        if force == 0:
            with O.max_points_per_node(3000):
                want = O.build_closed(0.5, bmin, bmax, x, y, z, rgb, threads=2)
            assert got == set(want.nodes)  # 60 000 duplicates end in a node at the resolution limit
        else:
            for c in range(8):
                if (force >> c) & 1 and f"r{c}" in got:
                    assert any(k.startswith(f"r{c}") and len(k) == 3 for k in got)  # split although small


def test_settle_and_climb_work_lists_partition_every_leaf_exactly_once():
    """K6 work lists (csrc/pcv_spec.cpp pcv_settle_items / pcv_climb_layout): the leaf-wise kernels rely on every sorted slot
    lying in exactly one settle item of its own leaf (<= kPcvSettleTile = 1 024 slots) and every climber record — every 8th point of a leaf
    whose node is not the root (generation.rs:222-238: `i % 8 == 0`), dense per leaf from climb_base — in exactly one
    climb item of its own leaf (<= 256 records)."""
    lib = pcv.load_library()
    f = lib.pcv_worklist_selftest
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 5 + [C.c_void_p] + [C.c_void_p] * 4
    rng = np.random.default_rng(5)
    special = [0, 1, 7, 8, 9, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 100_000, 100_003]
    count = np.array(special + list(rng.integers(0, 5000, 300)) + list(rng.integers(0, 200_000, 40)), dtype=np.uint32)
    rng.shuffle(count)
    nl = count.size
    lo = np.concatenate([[0], np.cumsum(count.astype(np.uint64))[:-1]]).astype(np.uint32)
    climbs = (rng.random(nl) < 0.9).astype(np.uint8)
    n = int(count.astype(np.uint64).sum())
    TILE = 1024  # kPcvSettleTile (csrc/pcv_spec.h)
    settle = np.zeros((n // TILE + nl + 1, 4), dtype=np.uint32)
    climb = np.zeros((n // 8 // 256 + nl + 1, 4), dtype=np.uint32)
    base = np.zeros(nl, dtype=np.uint32)
    ns, nc, total = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert f(lo.ctypes.data, count.ctypes.data, climbs.ctypes.data, nl, settle.ctypes.data, C.byref(ns), base.ctypes.data,
             climb.ctypes.data, C.byref(nc), C.byref(total)) == 0
    settle, climb = settle[:ns.value], climb[:nc.value]
    # settle: in slot order, back to back, one leaf each, <= TILE slots, nothing for empty leaves
    assert ns.value == int(np.sum((count.astype(np.int64) + TILE - 1) // TILE))
    assert np.all(settle[:, 2] > settle[:, 1]) and np.all(settle[:, 2] - settle[:, 1] <= TILE)
    nz = np.flatnonzero(count)
    assert settle[0, 1] == lo[nz[0]] and settle[-1, 2] == n and np.all(settle[1:, 1] == settle[:-1, 2])
    r = settle[:, 0]
    assert np.all(settle[:, 1] >= lo[r]) and np.all(settle[:, 2].astype(np.uint64) <= lo[r].astype(np.uint64) + count[r])
    # climbers: ceil(count / 8) per climbing leaf, dense in rank order
    k8 = np.where(climbs != 0, (count.astype(np.int64) + 7) // 8, 0)
    assert total.value == int(k8.sum())
    assert np.array_equal(base.astype(np.int64), np.concatenate([[0], np.cumsum(k8)[:-1]]))
    assert nc.value == int(np.sum((k8 + 255) // 256))
    if nc.value:
        rc_ = climb[:, 0]
        assert np.all(climb[:, 2] > climb[:, 1]) and np.all(climb[:, 2] - climb[:, 1] <= 256)
        assert climb[0, 1] == base[np.flatnonzero(k8)[0]] and climb[-1, 2] == total.value
        assert np.all(climb[1:, 1] == climb[:-1, 2])
        assert np.all(climb[:, 1] >= base[rc_]) and np.all(climb[:, 2].astype(np.int64) <= base[rc_].astype(np.int64) + k8[rc_])
        assert np.all(climbs[rc_] != 0)
        assert np.array_equal(climb[:, 3], base[rc_])  # pad = the leaf's first climber record
