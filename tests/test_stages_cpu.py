"""pcv_promote_assign (host-only stage entry point): the closed form of the every-8th promotion
(src/octree/generation.rs:195-253, 335-387) on a node table, against the oracle's finished octree: points kept per
node, and — through index colours — the exact node and slot every point ends in."""
import numpy as np

import oracle_lib as O
from point_cloud_viewer_amd import _lib as L
from point_cloud_viewer_amd import octree, synthetic


def _table_from_oracle(tree, keys_sorted, nlevels):
    """pcv_split_node table (breadth first, children consecutive in digit order) of an oracle tree."""
    names = sorted(tree.nodes, key=lambda k: (len(k), k))
    index = {k: i for i, k in enumerate(names)}
    nodes = (L.SplitNode * len(names))()
    for i, k in enumerate(names):
        level = len(k) - 1
        pfx = sum(int(d) << (3 * (21 - j)) for j, d in enumerate(k[1:], start=1))
        lo = int(np.searchsorted(keys_sorted, np.uint64(pfx), "left"))
        hi = int(np.searchsorted(keys_sorted, np.uint64(pfx + (1 << (3 * (21 - level))) - 1), "right")) if level else keys_sorted.size
        children = [k + str(c) for c in range(8) if k + str(c) in index]
        hi_lo = O.node_id_from_str(k)
        nodes[i].id_high, nodes[i].id_low = hi_lo
        nodes[i].first, nodes[i].count = lo, hi - lo
        nodes[i].level = level
        nodes[i].parent = index[k[:-1]] if level else 0xFFFFFFFF
        nodes[i].first_child = index[children[0]] if children else 0
        nodes[i].child_mask = sum(1 << int(c[-1]) for c in children)
        nodes[i].is_leaf = 0 if children else 1
    return names, nodes


def test_promote_assign_matches_the_oracle_octree():
    n, cap = 120_000, 900
    x, y, z, _, bmin, bmax = synthetic.gaussian_clusters(n, seed=14, num_clusters=4, extent=40.0, sigma_range=(0.01, 2.0))
    rgb = synthetic.index_colors(n)
    with O.max_points_per_node(cap):
        tree = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    ml, _, _ = O.level_table(bmin, bmax, 0.001)
    keys = O.chain_keys64(bmin, bmax, 0.001, min(ml, 21), x, y, z, threads=4)
    names, nodes = _table_from_oracle(tree, np.sort(keys), min(ml, 21))
    stream, kept, off, node_of, slot_in = octree.promote_assign(nodes, len(names), n, with_slots=True)
    assert kept.sum() == n
    for i, k in enumerate(names):
        assert kept[i] == tree.nodes[k]["num_points"], k
    # leaf-sorted order == stable sort of the input by the key prefix of its leaf; leaves are contiguous key ranges, so a
    # stable argsort of the full keys restricted to a leaf, re-sorted by input index, is that leaf's stream
    order = np.argsort(keys, kind="stable")
    pos = np.empty(n, dtype=np.int64)
    for i, k in enumerate(names):
        if nodes[i].is_leaf:
            lo, cnt = nodes[i].first, nodes[i].count
            members = np.sort(order[lo:lo + cnt])  # input order inside the leaf
            pos[lo:lo + cnt] = members
    for i, k in enumerate(names):
        c = np.frombuffer(tree.nodes[k]["rgb"], dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        want = (c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]  # input indices of the node's points, in file order
        sel = np.nonzero(node_of == i)[0]
        got = np.empty(len(sel), dtype=np.int64)
        got[slot_in[sel]] = pos[sel]
        assert np.array_equal(got, want), k


def test_promote_assign_rejects_a_broken_table():
    nodes = (L.SplitNode * 2)()
    nodes[0].is_leaf, nodes[0].child_mask, nodes[0].first_child, nodes[0].count = 0, 0b11, 1, 10  # two children, one entry
    nodes[1].is_leaf, nodes[1].count, nodes[1].parent = 1, 10, 0
    import pytest
    with pytest.raises(L.PcvError):
        octree.promote_assign(nodes, 2)
