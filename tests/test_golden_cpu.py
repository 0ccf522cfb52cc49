"""The oracle against the frozen fixtures of tests/golden/ (see make_golden.py for what they are and are not), both
formulations: closed form and the literal file-streaming restatement. For the reference's own unit-test cloud
(src/octree/tests.rs:18-46) the fixture must also carry the node counts derived from the reference in SURVEY §8c."""
import importlib.util
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
GOLDEN = json.load(open(os.path.join(HERE, "golden", "build_golden.json")))


@pytest.mark.parametrize("case", G.CASES, ids=[c[0] for c in G.CASES])
def test_oracle_matches_golden(case):
    x, y, z, rgb, inten, bmin, bmax, res, cap = G.make_case(case)
    want = GOLDEN[case[0]]
    assert want["points"] == x.size and want["bbox_min"] == [float(v) for v in bmin]
    with O.max_points_per_node(cap or 100_000):
        closed = O.build_closed(res, bmin, bmax, x, y, z, rgb, inten, threads=2)
        literal = O.build_literal(res, bmin, bmax, x, y, z, rgb, inten, threads=2) if x.size <= 50_000 else None
    assert G.digest(closed.nodes) == want["nodes"]
    if literal is not None:
        assert G.digest(literal.nodes) == want["nodes"]


def test_reference_unit_test_counts():
    nodes = GOLDEN["reference_unit_test_cloud"]["nodes"]
    assert {k: v[0] for k, v in nodes.items()} == {"r": 12501, "r0": 0, "r4": 87500}
    assert all(v[1] == 1 for v in nodes.values())  # both levels Uint8
