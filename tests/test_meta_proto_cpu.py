"""meta.pb written by the library, parsed by the REAL protobuf runtime against the reference's schema
(point_viewer_proto_rust/src/proto.proto:58-149; loader src/octree/mod.rs:156-215 unwraps `id` of every node)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import meta_proto  # noqa: E402


def test_schema_matches_the_reference_proto_file():
    compared = meta_proto.check_against_reference()
    assert compared is None or compared >= 25


def test_meta_pb_parses_with_the_protobuf_runtime(tmp_path):
    from point_cloud_viewer_amd import octree
    Meta = meta_proto.classes()["Meta"]
    nodes = [meta_proto.node_id(0, 0) + (12_501, 1),               # root: id is all zero, must still be present
             meta_proto.node_id(1, 0) + (0, 1),                    # a node without points (no files, still in meta)
             meta_proto.node_id(1, 4) + (87_500, 2),
             meta_proto.node_id(3, 0o705) + (7, 3),
             meta_proto.node_id(25, int("7" * 25, 8)) + (1 << 40, 4),  # index needs more than 64 bits -> high != level only
             meta_proto.node_id(40, int("1234567" * 5 + "01234", 8)) + (3, 1)]
    bmin, bmax = np.array([-200.0, -40.0, 0.0]), np.array([0.25, 1e7, 30.0])
    octree.write_meta(tmp_path, 0.001, bmin, bmax, nodes)
    raw = (tmp_path / "meta.pb").read_bytes()
    m = Meta.FromString(raw)
    assert m.version == 13
    assert m.HasField("bounding_box") and m.bounding_box.HasField("min") and m.bounding_box.HasField("max")
    assert [m.bounding_box.min.x, m.bounding_box.min.y, m.bounding_box.min.z] == bmin.tolist()
    assert [m.bounding_box.max.x, m.bounding_box.max.y, m.bounding_box.max.z] == bmax.tolist()
    assert not m.bounding_box.HasField("deprecated_min") and not m.bounding_box.HasField("deprecated_max")
    assert m.WhichOneof("data") == "octree" and m.octree.resolution == 0.001
    assert not m.octree.HasField("deprecated_bounding_box")
    assert m.deprecated_resolution == 0.0 and len(m.deprecated_nodes) == 0
    assert len(m.octree.nodes) == len(nodes)
    for got, (hi, lo, npts, enc) in zip(m.octree.nodes, nodes):
        assert got.HasField("id")            # octree/mod.rs:199 `node_proto.id.as_ref().unwrap()`
        assert got.position_encoding == enc and got.position_encoding != 0  # codec.rs:50-53 rejects INVALID
        assert (got.id.high, got.id.low, got.num_points) == (hi, lo, npts)
        assert got.id.deprecated_level == 0 and got.id.deprecated_index == 0
        assert got.id.high >> 56 == (hi >> 56)
    # nothing the schema does not know, and the canonical serialisation of the parsed message is the file itself
    from google.protobuf import unknown_fields
    for msg in [m, m.octree, m.bounding_box] + list(m.octree.nodes):
        assert len(unknown_fields.UnknownFieldSet(msg)) == 0
    assert m.SerializeToString(deterministic=True) == raw


def test_meta_pb_of_an_empty_octree(tmp_path):
    from point_cloud_viewer_amd import octree
    Meta = meta_proto.classes()["Meta"]
    octree.write_meta(tmp_path, 1.0, [0, 0, 0], [0, 0, 0], [])
    m = Meta.FromString((tmp_path / "meta.pb").read_bytes())
    assert m.version == 13 and m.WhichOneof("data") == "octree" and len(m.octree.nodes) == 0 and m.octree.resolution == 1.0
