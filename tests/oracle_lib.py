"""ctypes binding of the CPU ORACLE (oracle/libpcv_oracle.so).

Test infrastructure only: imported by tests/, by bench.py's cpu_baseline leg and by
__graft_entry__.smoke() as the checker. The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "libpcv_oracle.so")

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libpcv_oracle.so"])


def _load():
    if not os.path.exists(_SO):
        build_oracle()
    lib = C.CDLL(_SO)
    lib.pcvo_encode_coord.restype = C.c_uint64
    lib.pcvo_encode_coord.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    lib.pcvo_decode_coord.restype = C.c_double
    lib.pcvo_decode_coord.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_double]
    lib.pcvo_position_encoding.restype = C.c_int
    lib.pcvo_position_encoding.argtypes = [C.c_double, C.c_double]
    lib.pcvo_child_index.restype = C.c_int
    lib.pcvo_child_index.argtypes = [_dp, C.c_double, _dp]
    lib.pcvo_node_id_from_string.argtypes = [C.c_char_p, _u64p, _u64p]
    lib.pcvo_node_id_to_string.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
    lib.pcvo_node_id_parent.restype = C.c_int
    lib.pcvo_node_id_parent.argtypes = [C.c_uint64, C.c_uint64, _u64p, _u64p]
    lib.pcvo_node_id_child_index.restype = C.c_int
    lib.pcvo_node_id_child_index.argtypes = [C.c_uint64, C.c_uint64]
    lib.pcvo_node_id_child.argtypes = [C.c_uint64, C.c_uint64, C.c_int, _u64p, _u64p]
    lib.pcvo_find_bounding_cube.argtypes = [C.c_uint64, C.c_uint64, _dp, C.c_double, _dp, _dp]
    lib.pcvo_cube_bounding.argtypes = [_dp, _dp, _dp, _dp]
    lib.pcvo_level_table.restype = C.c_int
    lib.pcvo_level_table.argtypes = [_dp, _dp, C.c_double, C.c_int, _dp, _ip]
    lib.pcvo_chain_keys64.argtypes = [_dp, _dp, C.c_double, C.c_int, C.c_uint64, _dp, _dp, _dp, _u64p, C.c_int]
    lib.pcvo_aabb.argtypes = [C.c_uint64, _dp, _dp, _dp, _dp, _dp]
    lib.pcvo_build_literal_dir.restype = C.c_int
    lib.pcvo_build_literal_dir.argtypes = [C.c_char_p, C.c_double, _dp, _dp, C.c_uint64, _dp, _dp, _dp, _u8p, _fp,
                                           C.c_uint64, C.c_int]
    lib.pcvo_build_literal_mem.restype = C.c_void_p
    lib.pcvo_build_literal_mem.argtypes = [C.c_double, _dp, _dp, C.c_uint64, _dp, _dp, _dp, _u8p, _fp, C.c_uint64,
                                           C.c_int]
    lib.pcvo_build_closed.restype = C.c_void_p
    lib.pcvo_build_closed.argtypes = [C.c_double, _dp, _dp, C.c_uint64, _dp, _dp, _dp, _u8p, _fp, C.c_int]
    lib.pcvo_load_dir.restype = C.c_void_p
    lib.pcvo_load_dir.argtypes = [C.c_char_p]
    lib.pcvo_result_error.restype = C.c_char_p
    lib.pcvo_result_error.argtypes = [C.c_void_p]
    lib.pcvo_result_version.restype = C.c_int
    lib.pcvo_result_version.argtypes = [C.c_void_p]
    lib.pcvo_result_resolution.restype = C.c_double
    lib.pcvo_result_resolution.argtypes = [C.c_void_p]
    lib.pcvo_result_bbox.argtypes = [C.c_void_p, _dp, _dp]
    lib.pcvo_result_num_nodes.restype = C.c_uint64
    lib.pcvo_result_num_nodes.argtypes = [C.c_void_p]
    lib.pcvo_result_node.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, _i64p, _ip, _ip, _ip]
    lib.pcvo_result_node_data.restype = C.POINTER(C.c_uint8)
    lib.pcvo_result_node_data.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _u64p]
    lib.pcvo_result_free.argtypes = [C.c_void_p]
    lib.pcvo_meta_encode.restype = C.c_uint64
    lib.pcvo_meta_encode.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_uint64, _u64p, _u64p, _i64p, _ip, _u8p,
                                     C.c_uint64]
    lib.pcvo_num_procs.restype = C.c_int
    lib.pcvo_set_max_points_per_node.argtypes = [C.c_int64]
    lib.pcvo_get_max_points_per_node.restype = C.c_int64
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _vec3(v):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
    assert a.shape == (3,)
    return a


def node_id_str(hi, lo):
    buf = C.create_string_buffer(64)
    lib().pcvo_node_id_to_string(int(hi), int(lo), buf, 64)
    return buf.value.decode()


def node_id_from_str(s):
    hi, lo = C.c_uint64(), C.c_uint64()
    lib().pcvo_node_id_from_string(s.encode(), C.byref(hi), C.byref(lo))
    return hi.value, lo.value


class Octree:
    """A finished octree as plain python: {node name: (num_points, encoding, xyz bytes, rgb bytes, intensity bytes)}."""

    def __init__(self, handle):
        L = lib()
        err = L.pcvo_result_error(handle)
        if err:
            msg = err.decode()
            L.pcvo_result_free(handle)
            raise RuntimeError("oracle: " + msg)
        self.version = L.pcvo_result_version(handle)
        self.resolution = L.pcvo_result_resolution(handle)
        bmin = np.zeros(3)
        bmax = np.zeros(3)
        L.pcvo_result_bbox(handle, _d(bmin), _d(bmax))
        self.bbox_min, self.bbox_max = bmin, bmax
        self.nodes = {}
        self.order = []
        n = L.pcvo_result_num_nodes(handle)
        for i in range(n):
            hi, lo = C.c_uint64(), C.c_uint64()
            npnts = C.c_int64()
            enc, lvl, has = C.c_int(), C.c_int(), C.c_int()
            L.pcvo_result_node(handle, i, C.byref(hi), C.byref(lo), C.byref(npnts), C.byref(enc), C.byref(lvl),
                               C.byref(has))
            blobs = []
            for which in range(3):
                ln = C.c_uint64()
                p = L.pcvo_result_node_data(handle, i, which, C.byref(ln))
                blobs.append(C.string_at(p, ln.value) if ln.value else b"")
            name = node_id_str(hi.value, lo.value)
            self.order.append(name)
            self.nodes[name] = dict(id=(hi.value, lo.value), num_points=npnts.value, encoding=enc.value,
                                    level=lvl.value, files=has.value, xyz=blobs[0], rgb=blobs[1],
                                    intensity=blobs[2])
        L.pcvo_result_free(handle)

    def total_points(self):
        return sum(n["num_points"] for n in self.nodes.values())


def _pts(x, y, z, rgb, intensity):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1)
    assert rgb.size == 3 * x.size
    ip = None
    if intensity is not None:
        intensity = np.ascontiguousarray(intensity, dtype=np.float32)
        ip = intensity.ctypes.data_as(_fp)
    return x, y, z, rgb, intensity, ip


def build_literal(resolution, bmin, bmax, x, y, z, rgb, intensity=None, batch_size=0, threads=1):
    x, y, z, rgb, intensity, ip = _pts(x, y, z, rgb, intensity)
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    h = lib().pcvo_build_literal_mem(resolution, _d(bmin), _d(bmax), x.size, _d(x), _d(y), _d(z),
                                     rgb.ctypes.data_as(_u8p), ip, batch_size, threads)
    return Octree(h)


def build_literal_dir(path, resolution, bmin, bmax, x, y, z, rgb, intensity=None, batch_size=0, threads=1):
    x, y, z, rgb, intensity, ip = _pts(x, y, z, rgb, intensity)
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    return lib().pcvo_build_literal_dir(str(path).encode(), resolution, _d(bmin), _d(bmax), x.size, _d(x), _d(y),
                                        _d(z), rgb.ctypes.data_as(_u8p), ip, batch_size, threads)


def build_closed(resolution, bmin, bmax, x, y, z, rgb, intensity=None, threads=1):
    x, y, z, rgb, intensity, ip = _pts(x, y, z, rgb, intensity)
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    h = lib().pcvo_build_closed(resolution, _d(bmin), _d(bmax), x.size, _d(x), _d(y), _d(z),
                                rgb.ctypes.data_as(_u8p), ip, threads)
    return Octree(h)


def load_dir(path):
    return Octree(lib().pcvo_load_dir(str(path).encode()))


def level_table(bmin, bmax, resolution, cap=40):
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    edge = np.zeros(cap + 2)
    enc = np.zeros(cap + 2, dtype=np.int32)
    ml = lib().pcvo_level_table(_d(bmin), _d(bmax), resolution, cap, _d(edge), enc.ctypes.data_as(_ip))
    return ml, edge[:ml + 1].copy(), enc[:ml + 1].copy()


def chain_keys64(bmin, bmax, resolution, nlevels, x, y, z, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    keys = np.zeros(x.size, dtype=np.uint64)
    lib().pcvo_chain_keys64(_d(bmin), _d(bmax), resolution, nlevels, x.size, _d(x), _d(y), _d(z),
                            keys.ctypes.data_as(_u64p), threads)
    return keys


def aabb(x, y, z):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    bmin, bmax = np.zeros(3), np.zeros(3)
    lib().pcvo_aabb(x.size, _d(x), _d(y), _d(z), _d(bmin), _d(bmax))
    return bmin, bmax


class max_points_per_node:
    """Context manager: temporarily lower the reference's MAX_POINTS_PER_NODE (generation.rs:37) so small
    clouds build deep trees. Test knob only."""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = lib().pcvo_get_max_points_per_node()
        lib().pcvo_set_max_points_per_node(self.v)

    def __exit__(self, *a):
        lib().pcvo_set_max_points_per_node(self.old)


def num_procs():
    return lib().pcvo_num_procs()


def compare_octrees(a, b, check_bytes=True):
    """Return a list of human-readable differences between two Octree objects (empty == identical)."""
    diffs = []
    if a.version != b.version:
        diffs.append(f"version {a.version} != {b.version}")
    if a.resolution != b.resolution:
        diffs.append(f"resolution {a.resolution} != {b.resolution}")
    if not (np.array_equal(a.bbox_min, b.bbox_min) and np.array_equal(a.bbox_max, b.bbox_max)):
        diffs.append("bounding boxes differ")
    ka, kb = set(a.nodes), set(b.nodes)
    for k in sorted(ka - kb):
        diffs.append(f"node {k} only in A")
    for k in sorted(kb - ka):
        diffs.append(f"node {k} only in B")
    for k in sorted(ka & kb):
        na, nb = a.nodes[k], b.nodes[k]
        for f in ("id", "num_points", "encoding", "files"):
            if na[f] != nb[f]:
                diffs.append(f"node {k}: {f} {na[f]} != {nb[f]}")
        if check_bytes:
            for f in ("xyz", "rgb", "intensity"):
                if na[f] != nb[f]:
                    x = np.frombuffer(na[f], dtype=np.uint8)
                    y = np.frombuffer(nb[f], dtype=np.uint8)
                    if x.size != y.size:
                        diffs.append(f"node {k}: {f} size {x.size} != {y.size}")
                    else:
                        bad = np.nonzero(x != y)[0]
                        diffs.append(f"node {k}: {f} differs in {bad.size} bytes, first at {bad[0]}")
        if len(diffs) > 50:
            diffs.append("... (truncated)")
            break
    return diffs
