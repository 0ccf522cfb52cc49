"""Stage-level parity of the HIP kernels against the CPU oracle (all through the C ABI)."""
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


def _clouds():
    x, y, z, rgb, bmin, bmax = synthetic.uniform_ecef(200_000)
    yield "uniform_ecef", x, y, z, bmin, bmax, 0.001
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(300_000, seed=4, num_clusters=5, extent=1000.0,
                                                           sigma_range=(0.01, 20.0))
    yield "clusters", x, y, z, bmin, bmax, 0.001
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(100_000, seed=6, num_clusters=8, extent=30000.0,
                                                           sigma_range=(1.0, 500.0), offset=(-2.7e6, -4.3e6, 3.8e6))
    yield "ecef_f64_levels", x, y, z, bmin, bmax, 0.001  # root edge > 16.7 km -> Float64 levels
    x, y, z, rgb, bmin, bmax, res = synthetic.reference_unit_test_cloud()
    yield "reference_unit_test", x, y, z, bmin, bmax, res


def test_aabb_reduce(ctx):
    for name, x, y, z, *_ in _clouds():
        bmin, bmax = ctx.aabb_reduce(x, y, z)
        omin, omax = O.aabb(x, y, z)
        assert np.array_equal(bmin, omin) and np.array_equal(bmax, omax), name
    bmin, bmax = ctx.aabb_reduce(np.zeros(0), np.zeros(0), np.zeros(0))
    assert not bmin.any() and not bmax.any()  # Aabb::zero()
    one = ctx.aabb_reduce(np.array([3.0]), np.array([-4.0]), np.array([5.5]))
    assert one[0].tolist() == [3.0, -4.0, 5.5] and one[1].tolist() == [3.0, -4.0, 5.5]


def test_aabb_reduce_odd_sizes_and_unaligned_views(ctx):
    import torch
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 63, 64, 65, 511, 4097, 100_003):
        p = rng.normal(size=(3, n + 1)) * 1e3
        bmin, bmax = ctx.aabb_reduce(p[0, :n], p[1, :n], p[2, :n])
        assert np.array_equal(bmin, p[:, :n].min(axis=1)) and np.array_equal(bmax, p[:, :n].max(axis=1))
        t = torch.from_numpy(p).cuda()
        bmin, bmax = ctx.aabb_reduce(t[0, 1:], t[1, 1:], t[2, 1:])  # 8-byte-aligned device views
        assert np.array_equal(bmin, p[:, 1:].min(axis=1)) and np.array_equal(bmax, p[:, 1:].max(axis=1))


def test_chain_keys_bit_exact(ctx):
    for name, x, y, z, bmin, bmax, res in _clouds():
        ml, _, _ = pcv.level_table(bmin, bmax, res)
        nl = min(ml, 21)
        keys = ctx.chain_keys(res, pcv.Aabb(bmin, bmax), x, y, z)
        want = O.chain_keys64(bmin, bmax, res, nl, x, y, z, threads=4)
        bad = np.nonzero(keys != want)[0]
        assert bad.size == 0, f"{name}: {bad.size} keys differ, first {bad[:5]}"


def test_chain_keys_special_values(ctx):
    # points on / outside the (loose) bounding box, exact cube centres, signed zeros
    bmin, bmax = np.array([-8.0, -8.0, -8.0]), np.array([8.0, 8.0, 8.0])
    g = np.array([-9.0, -8.0, -4.0, -0.0, 0.0, 1e-300, 4.0, 7.999999999, 8.0, 12.5, 2.0 ** -30, -2.0 ** -30,
                  float("nan"), float("inf"), -float("inf"), 1e200, -1e200, 5e-324])
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g)]
    for res in (1.0, 0.001, 1e-7):
        ml, _, _ = pcv.level_table(bmin, bmax, res)
        keys = ctx.chain_keys(res, pcv.Aabb(bmin, bmax), x, y, z)
        want = O.chain_keys64(bmin, bmax, res, min(ml, 21), x, y, z)
        assert np.array_equal(keys, want), res


def test_sort_keys64(ctx):
    rng = np.random.default_rng(1)
    for n in (0, 1, 5, 4095, 4096, 4097, 70_001, 1_000_003):
        keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
        if n > 10:
            keys[::7] = keys[3]  # heavy duplicates
        got = ctx.sort_keys64(keys.copy(), 0, 63)
        assert np.array_equal(got, np.sort(keys)), n
    # partial bit ranges sort on those bits only, stably: equal masked keys keep their input order
    keys = rng.integers(0, 2 ** 63, 200_000, dtype=np.uint64)
    got = ctx.sort_keys64(keys.copy(), 12, 33)
    m = (keys >> np.uint64(12)) & np.uint64((1 << 21) - 1)
    assert np.array_equal(got, keys[np.argsort(m, kind="stable")])


def test_sort_keys64_top_bit_ranges(ctx):
    """Key sorts over the top 25-42 bits of the word, like the path keys of the sample (30-42 significant bits, noise
    below): ragged sizes around the tiles and chunks, skewed upper digits, long runs of equal keys — stable."""
    rng = np.random.default_rng(11)
    for n, bits in ((1, 33), (63, 33), (8191, 33), (8192, 30), (8193, 36), (16_385, 39), (100_003, 42), (3_125_000, 33),
                    (1_000_003, 25), ((1 << 24), 33)):
        lo = 63 - bits
        body = rng.integers(0, 2 ** bits, n, dtype=np.uint64)
        body[: n // 3] &= np.uint64((1 << (bits - 9)) - 1)  # a third of the keys share the upper digit
        keys = (body << np.uint64(lo)) | rng.integers(0, 2 ** lo, n, dtype=np.uint64)  # noise below the sorted bits
        if n > 100:
            keys[::5] = keys[7]
        got = ctx.sort_keys64(keys.copy(), lo, 63)
        order = np.argsort(keys >> np.uint64(lo), kind="stable")
        assert np.array_equal(got, keys[order]), (n, bits)


def test_sort_keys32(ctx):
    """The 32-bit key sort the build uses when ten levels suffice: skewed digits (few distinct values in the upper
    bytes, like path keys), ragged sizes, partial bit ranges."""
    rng = np.random.default_rng(4)
    for n in (0, 1, 63, 4095, 4097, 99_999, 2_000_003):
        hi = rng.integers(0, 24, n, dtype=np.uint64) << np.uint64(24)  # 24 populated top buckets
        keys = (hi | rng.integers(0, 2 ** 24, n, dtype=np.uint64)).astype(np.uint32)
        if n > 100:
            keys[::5] = keys[7]
        assert np.array_equal(ctx.sort_keys32(keys.copy(), 0, 30), np.sort(keys)), n
    keys = rng.integers(0, 2 ** 30, 300_000, dtype=np.uint64).astype(np.uint32)
    got = ctx.sort_keys32(keys.copy(), 6, 30)  # the build sorts bits [3 * (10 - levels), 30)
    assert np.array_equal(got, keys[np.argsort(keys >> np.uint32(6), kind="stable")])


def test_sort_pairs32_is_stable(ctx):
    rng = np.random.default_rng(2)
    for n, bits in ((1, 1), (1000, 3), (123_457, 11), (1_000_000, 14), (300_000, 32)):
        keys = rng.integers(0, 2 ** bits, n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k, v = ctx.sort_pairs32(keys.copy(), vals.copy(), 0, bits)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order]), (n, bits)


def test_exact_constant_divisor_division_selftest(ctx):
    """pcv_div_const / pcv_div_code (Markstein) == IEEE division, bit for bit: every u8/u16 code exhaustively, and
    4 M pseudo-random numerators (any exponent, boundary-straddling quotients, inf/nan/denormals) per divisor for
    the edges of typical level tables plus awkward divisors."""
    divisors = []
    for root in (1000.1234567, 283.0, 200.0, 1.0, 6378137.0 * 0.37, 30000.0 / 7.0, 0.0123):
        e = root
        for _ in range(22):
            divisors.append(e)
            e /= 2.0
    divisors += [3.0, 7.0, 1.0 / 3.0, 0.1, 1e-30, 1e30, 1e-40, 1e200, 5e-324, 2.0 ** 52 - 1, 1.9999999999999998,
                 1.0000000000000002]
    assert ctx.selftest_division(divisors, 1 << 22) == 0


def test_node_split_promote_assign_gather_encode_stage_by_stage(ctx):
    """The build stage by stage through the C ABI: pcv_chain_keys -> pcv_sort_keys64 -> pcv_node_split ->
    pcv_promote_assign -> pcv_gather_encode, every stage against the oracle (generation.rs:58-253)."""
    from point_cloud_viewer_amd import octree
    n, cap = 250_000, 1500
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=23, num_clusters=5, extent=90.0, sigma_range=(0.01, 4.0))
    inten = (np.arange(n) % 101).astype(np.float32)
    box = pcv.Aabb(bmin, bmax)
    with O.max_points_per_node(cap):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    keys = ctx.chain_keys(0.001, box, x, y, z)
    assert np.array_equal(keys, O.chain_keys64(bmin, bmax, 0.001, 21, x, y, z, threads=4))
    skeys = ctx.sort_keys64(keys.copy(), 0, 63)
    assert np.array_equal(skeys, np.sort(keys))
    nodes, m = ctx.node_split(0.001, box, skeys, max_points_per_node=cap)
    names = [pcv.node_name(nodes[i].id_high, nodes[i].id_low) for i in range(m)]
    assert set(names) == set(want.nodes) and len(names) == len(set(names))
    for i, k in enumerate(names):
        children = [c for c in want.nodes if len(c) == len(k) + 1 and c.startswith(k)]
        assert bool(nodes[i].is_leaf) == (not children), k
        assert nodes[i].level == len(k) - 1
        if children:
            got = [names[nodes[i].first_child + c] for c in range(bin(nodes[i].child_mask).count("1"))]
            assert got == sorted(children), k  # consecutive, digit order
            assert all(nodes[nodes[i].first_child + c].parent == i for c in range(len(got)))
        pfx = sum(int(d) << (3 * (21 - j)) for j, d in enumerate(k[1:], start=1))
        lo = int(np.searchsorted(skeys, np.uint64(pfx), "left"))
        hi = int(np.searchsorted(skeys, np.uint64(pfx + (1 << (3 * (21 - nodes[i].level))) - 1), "right")) if nodes[i].level else n
        assert (nodes[i].first, nodes[i].count) == (lo, hi - lo), k
    stream, kept, off = octree.promote_assign(nodes, m)
    for i, k in enumerate(names):
        assert kept[i] == want.nodes[k]["num_points"], k
    tree = ctx.gather_encode(0.001, box, x, y, z, rgb, nodes, m, intensity=inten, max_points_per_node=cap)
    got = tree.to_dict()
    assert set(got) == set(want.nodes)
    for k, nd in want.nodes.items():
        g = got[k]
        assert (g["num_points"], g["xyz"], g["rgb"], g["intensity"]) == (nd["num_points"], nd["xyz"], nd["rgb"], nd["intensity"]), k
    # a topology that does not belong to the points is refused, not encoded
    with pytest.raises(pcv.PcvError):
        ctx.gather_encode(0.001, box, x[:1000], y[:1000], z[:1000], rgb[:1000], nodes, m, max_points_per_node=cap)
    # depth overflow of the stage call (keys end at level 21)
    dup = np.zeros(5000)
    kz = ctx.chain_keys(1e-9, pcv.Aabb([0, 0, 0], [1, 1, 1]), dup, dup, dup)
    with pytest.raises(pcv.PcvError) as e:
        ctx.node_split(1e-9, pcv.Aabb([0, 0, 0], [1, 1, 1]), np.sort(kz), max_points_per_node=100)
    assert e.value.code == pcv.PCV_E_DEPTH
