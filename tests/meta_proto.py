"""The reference's `Meta` schema (point_viewer_proto_rust/src/proto.proto:27-149) as a protobuf descriptor built at
run time with the REAL protobuf runtime — an independent reader/writer for meta.pb (the library's own writer and
reader are hand-rolled). Field numbers are restated here; `check_against_reference()` re-derives them from the
reference's proto.proto when /root/reference is present (build container only)."""
import os
import re

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto
_SCALAR = {"double": _T.TYPE_DOUBLE, "float": _T.TYPE_FLOAT, "int32": _T.TYPE_INT32, "int64": _T.TYPE_INT64,
           "uint64": _T.TYPE_UINT64, "string": _T.TYPE_STRING}

# message -> [(name, number, type, repeated)]; types: scalar name, or message / enum name
SCHEMA = {
    "Vector3f": [("x", 1, "float", False), ("y", 2, "float", False), ("z", 3, "float", False)],
    "Vector3d": [("x", 1, "double", False), ("y", 2, "double", False), ("z", 3, "double", False)],
    "AxisAlignedCuboid": [("min", 3, "Vector3d", False), ("max", 4, "Vector3d", False),
                          ("deprecated_min", 1, "Vector3f", False), ("deprecated_max", 2, "Vector3f", False)],
    "NodeId": [("high", 3, "uint64", False), ("low", 4, "uint64", False), ("deprecated_level", 1, "int32", False),
               ("deprecated_index", 2, "int64", False)],
    "OctreeNode": [("position_encoding", 2, "PositionEncoding", False), ("num_points", 3, "int64", False),
                   ("id", 4, "NodeId", False)],
    "OctreeMeta": [("resolution", 2, "double", False), ("nodes", 3, "OctreeNode", True),
                   ("deprecated_bounding_box", 1, "AxisAlignedCuboid", False)],
    "S2Cell": [("id", 1, "uint64", False), ("num_points", 2, "uint64", False)],
    "S2Meta": [("cells", 1, "S2Cell", True)],
    "Meta": [("version", 1, "int32", False), ("bounding_box", 4, "AxisAlignedCuboid", False),
             ("octree", 6, "OctreeMeta", False), ("s2", 7, "S2Meta", False),
             ("deprecated_resolution", 3, "double", False), ("deprecated_nodes", 5, "OctreeNode", True)],
}
ENUMS = {"PositionEncoding": [("INVALID", 0), ("Uint8", 1), ("Uint16", 2), ("Float32", 3), ("Float64", 4)]}
ONEOF = {"Meta": ("data", ("octree", "s2"))}

_classes = None


def classes():
    """{message name: generated class} for package point_viewer.proto."""
    global _classes
    if _classes is not None:
        return _classes
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "pcv_test_meta.proto"
    fd.package = "point_viewer.proto"
    fd.syntax = "proto3"
    for ename, values in ENUMS.items():
        e = fd.enum_type.add()
        e.name = ename
        for vname, num in values:
            v = e.value.add()
            v.name, v.number = vname, num
    for mname, fields in SCHEMA.items():
        m = fd.message_type.add()
        m.name = mname
        if mname in ONEOF:
            m.oneof_decl.add().name = ONEOF[mname][0]
        for fname, num, typ, rep in fields:
            f = m.field.add()
            f.name, f.number = fname, num
            f.label = _T.LABEL_REPEATED if rep else _T.LABEL_OPTIONAL
            if typ in _SCALAR:
                f.type = _SCALAR[typ]
            elif typ in ENUMS:
                f.type, f.type_name = _T.TYPE_ENUM, ".point_viewer.proto." + typ
            else:
                f.type, f.type_name = _T.TYPE_MESSAGE, ".point_viewer.proto." + typ
            if mname in ONEOF and fname in ONEOF[mname][1]:
                f.oneof_index = 0
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    _classes = {m: message_factory.GetMessageClass(pool.FindMessageTypeByName("point_viewer.proto." + m)) for m in SCHEMA}
    return _classes


def check_against_reference(path="/root/reference/point_viewer_proto_rust/src/proto.proto"):
    """Field numbers / types / labels of SCHEMA and ENUMS against the reference's own proto file. Returns the number of
    fields compared, or None when the reference checkout is not there (GPU box)."""
    if not os.path.exists(path):
        return None
    text = re.sub(r"//[^\n]*", "", open(path).read())
    compared = 0
    for mname, fields in SCHEMA.items():
        body = re.search(r"message\s+%s\s*\{(.*?)\n\}" % mname, text, re.S).group(1)
        found = {}
        for rep, typ, name, num in re.findall(r"(repeated\s+)?([\w.]+)\s+(\w+)\s*=\s*(\d+)\s*;", body):
            found[name] = (int(num), typ, bool(rep))
        for fname, num, typ, rep in fields:
            assert found[fname] == (num, typ, rep), (mname, fname, found[fname], (num, typ, rep))
            compared += 1
        if mname != "S2Meta":  # (attributes of S2Meta are not part of the octree path)
            assert set(found) == {f[0] for f in fields}, (mname, sorted(found))
    for ename, values in ENUMS.items():
        body = re.search(r"enum\s+%s\s*\{(.*?)\}" % ename, text, re.S).group(1)
        assert [(n, int(v)) for n, v in re.findall(r"(\w+)\s*=\s*(\d+)\s*;", body)] == values
    return compared


def node_id(level, index):
    """(high, low) of NodeId u128 = level << 120 | index (src/octree/node.rs:101-111)."""
    v = (level << 120) | index
    return v >> 64, v & ((1 << 64) - 1)
