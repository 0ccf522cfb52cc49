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
    # query path
    lib.pcvo_mat4_try_inverse.restype = C.c_int
    lib.pcvo_mat4_try_inverse.argtypes = [_dp, _dp]
    lib.pcvo_perspective_new.argtypes = [C.c_double] * 6 + [_dp]
    lib.pcvo_perspective_inverse.argtypes = [_dp, _dp]
    lib.pcvo_perspective3_new.argtypes = [C.c_double] * 4 + [_dp]
    lib.pcvo_frustum_new.argtypes = [_dp, _dp, _dp, _dp]
    lib.pcvo_iso_transform_points.argtypes = [_dp, C.c_uint64, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.pcvo_cached_axes.restype = C.c_int
    lib.pcvo_cached_axes.argtypes = [C.c_int, _dp, _dp, _dp]
    lib.pcvo_cull_cubes.restype = C.c_int
    lib.pcvo_cull_cubes.argtypes = [C.c_int, _dp, C.c_uint64, _dp, _u8p, _dp]
    lib.pcvo_intersect_shapes.restype = C.c_int
    lib.pcvo_intersect_shapes.argtypes = [C.c_int, _dp, C.c_int, _dp]
    lib.pcvo_sat_raw.restype = C.c_int
    lib.pcvo_sat_raw.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, C.c_int]
    lib.pcvo_get_visible_nodes.restype = C.c_int64
    lib.pcvo_get_visible_nodes.argtypes = [_dp, _dp, C.c_uint64, _u64p, _u64p, _i64p, _dp, _u64p, _u64p]
    lib.pcvo_nodes_in_location.restype = C.c_int64
    lib.pcvo_nodes_in_location.argtypes = [_dp, _dp, C.c_uint64, _u64p, _u64p, _i64p, C.c_int, _dp, _u64p, _u64p]
    lib.pcvo_cull_points.argtypes = [C.c_int, _dp, C.c_uint64, _dp, _dp, _dp, _fp, _dp, _u8p]
    lib.pcvo_decode_positions.argtypes = [C.c_int, _dp, C.c_double, C.c_uint64, _u8p, _dp, _dp, _dp]
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


def find_bounding_cube(hi, lo, root_min, root_edge):
    """NodeId::find_bounding_cube (node.rs:157-172): (min xyz array, edge)."""
    out = np.zeros(3)
    edge = C.c_double()
    lib().pcvo_find_bounding_cube(int(hi), int(lo), _d(_vec3(root_min)), float(root_edge), _d(out), C.byref(edge))
    return out, edge.value


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


def node_digest(ptr, length):
    """blake2b-128 of `length` bytes at address `ptr` (no copy)."""
    import hashlib
    if not length:
        return hashlib.blake2b(b"", digest_size=16).hexdigest()
    return hashlib.blake2b((C.c_uint8 * length).from_address(ptr), digest_size=16).hexdigest()


def build_closed_digests(resolution, bmin, bmax, x, y, z, rgb, intensity=None, threads=1):
    """Closed-form oracle build of a LARGE cloud without copying node bytes into python: returns
    ({node name: (num_points, encoding, digest xyz, digest rgb, digest intensity)}, stats) with
    stats = dict(max_abs_position_error=..., total_points=...). Used by `bench.py --verify` at BASELINE sizes."""
    x, y, z, rgb, intensity, ip = _pts(x, y, z, rgb, intensity)
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    L = lib()
    L.pcvo_result_max_abs_position_error.restype = C.c_double
    L.pcvo_result_max_abs_position_error.argtypes = [C.c_void_p]
    h = L.pcvo_build_closed(resolution, _d(bmin), _d(bmax), x.size, _d(x), _d(y), _d(z), rgb.ctypes.data_as(_u8p), ip,
                            threads)
    try:
        err = L.pcvo_result_error(h)
        if err:
            raise RuntimeError("oracle: " + err.decode())
        out, total = {}, 0
        for i in range(L.pcvo_result_num_nodes(h)):
            hi, lo, npnts = C.c_uint64(), C.c_uint64(), C.c_int64()
            enc, lvl, has = C.c_int(), C.c_int(), C.c_int()
            L.pcvo_result_node(h, i, C.byref(hi), C.byref(lo), C.byref(npnts), C.byref(enc), C.byref(lvl), C.byref(has))
            dig = []
            for which in range(3):
                ln = C.c_uint64()
                ptr = L.pcvo_result_node_data(h, i, which, C.byref(ln))
                dig.append(node_digest(C.cast(ptr, C.c_void_p).value or 0, ln.value))
            out[node_id_str(hi.value, lo.value)] = (npnts.value, enc.value, dig[0], dig[1], dig[2])
            total += npnts.value
        return out, dict(max_abs_position_error=L.pcvo_result_max_abs_position_error(h), total_points=total)
    finally:
        L.pcvo_result_free(h)


def chain_state1(bmin, bmax, resolution, x, y, z):
    """Level-1 chain state of raw points: (octant uint8, cx, cy, cz uint32 raw level-1 codes)."""
    x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    oct_ = np.zeros(x.size, dtype=np.uint8)
    c = [np.zeros(x.size, dtype=np.uint32) for _ in range(3)]
    f = lib().pcvo_chain_state1
    f.restype = None
    f.argtypes = [_dp, _dp, C.c_double, C.c_uint64, _dp, _dp, _dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(_d(bmin), _d(bmax), resolution, x.size, _d(x), _d(y), _d(z), oct_.ctypes.data, c[0].ctypes.data, c[1].ctypes.data,
      c[2].ctypes.data)
    return oct_, c[0], c[1], c[2]


def build_closed_shard(resolution, bmin, bmax, x, y, z, rgb, intensity=None, threads=1, force_mask=0, layout=None,
                       routed=None):
    """One rank's share of a multi-rank build. Returns (Octree, streams[73]) — streams = 8 level-1 + 64 level-2 stream
    lengths + the local level-1 split mask. layout = array of 1 + 8 + 8 + 64 u64 (root_points, l1_stream, l1_offset,
    l2_offset) or None."""
    ro = None
    if routed is not None:  # (octant, cx, cy, cz) instead of coordinates
        ro = [np.ascontiguousarray(routed[0], dtype=np.uint8)] + [np.ascontiguousarray(a, dtype=np.uint32) for a in routed[1:]]
        x = y = z = np.zeros(ro[0].size)
    x, y, z, rgb, intensity, ip = _pts(x, y, z, rgb, intensity)
    bmin, bmax = _vec3(bmin), _vec3(bmax)
    streams = np.zeros(73, dtype=np.uint64)
    lay = None if layout is None else np.ascontiguousarray(layout, dtype=np.uint64)
    f = lib().pcvo_build_closed_shard
    f.restype = C.c_void_p
    f.argtypes = [C.c_double, _dp, _dp, C.c_uint64, _dp, _dp, _dp, _u8p, _fp, C.c_int, C.c_uint, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rp = [None] * 4 if ro is None else [a.ctypes.data for a in ro]
    h = f(resolution, _d(bmin), _d(bmax), x.size, _d(x), _d(y), _d(z), rgb.ctypes.data_as(_u8p), ip, threads,
          int(force_mask), None if lay is None else lay.ctypes.data, streams.ctypes.data, *rp)
    return Octree(h), streams


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


# ---- query path ---------------------------------------------------------------------------------------
SHAPE_ALL, SHAPE_AABB, SHAPE_FRUSTUM, SHAPE_OBB, SHAPE_FRUSTUM2 = 0, 1, 2, 3, 4
REL_IN, REL_CROSS, REL_OUT = 0, 1, 2


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())


def quat_from_axis_angle(axis, angle):
    """nalgebra UnitQuaternion::from_axis_angle (axis must be unit): (axis * sin(a/2), cos(a/2)) as i,j,k,w."""
    import math
    s, c = math.sin(angle / 2.0), math.cos(angle / 2.0)
    return [axis[0] * s, axis[1] * s, axis[2] * s, c]


def perspective_new(left, right, bottom, top, near, far):
    m = np.zeros(16)
    lib().pcvo_perspective_new(left, right, bottom, top, near, far, _d(m))
    return m


def perspective3_new(aspect, fovy, near, far):
    m = np.zeros(16)
    lib().pcvo_perspective3_new(aspect, fovy, near, far, _d(m))
    return m


def perspective_inverse(p):
    m = np.zeros(16)
    lib().pcvo_perspective_inverse(_d(_f64(p)), _d(m))
    return m


def mat4_try_inverse(m):
    out = np.zeros(16)
    ok = lib().pcvo_mat4_try_inverse(_d(_f64(m)), _d(out))
    return out if ok else None


def frustum_new(translation, quat, perspective):
    """Frustum::new(query_from_eye, clip_from_eye) -> (clip_from_query, query_from_clip), column-major 16 each."""
    iso = _f64(list(translation) + list(quat))
    c, q = np.zeros(16), np.zeros(16)
    lib().pcvo_frustum_new(_d(iso), _d(_f64(perspective)), _d(c), _d(q))
    return c, q


def cached_axes(kind, params):
    corners, axes = np.zeros(24), np.zeros(78)
    n = lib().pcvo_cached_axes(kind, _d(_f64(params)), _d(corners), _d(axes))
    if n < 0:
        return None
    return corners.reshape(8, 3), axes[:3 * n].reshape(n, 3)


def cull_cubes(kind, params, cubes4, with_sizes=False):
    cubes4 = _f64(cubes4)
    m = cubes4.size // 4
    rel = np.zeros(m, dtype=np.uint8)
    sizes = np.zeros(m) if with_sizes else None
    rc = lib().pcvo_cull_cubes(kind, _d(_f64(params)), m, _d(cubes4), rel.ctypes.data_as(_u8p),
                               _d(sizes) if with_sizes else None)
    assert rc == 0
    return (rel, sizes) if with_sizes else rel


def intersect_shapes(kind_a, pa, kind_b, pb):
    return lib().pcvo_intersect_shapes(kind_a, _d(_f64(pa)), kind_b, _d(_f64(pb)))


def _tree_arrays(nodes):
    """nodes: dict name -> dict(id=(hi,lo), num_points=..)"""
    names = list(nodes)
    hi = np.array([nodes[k]["id"][0] for k in names], dtype=np.uint64)
    lo = np.array([nodes[k]["id"][1] for k in names], dtype=np.uint64)
    npnts = np.array([nodes[k]["num_points"] for k in names], dtype=np.int64)
    return hi, lo, npnts


def get_visible_nodes(bmin, bmax, nodes, matrix):
    hi, lo, npnts = _tree_arrays(nodes)
    ohi, olo = np.zeros(hi.size, dtype=np.uint64), np.zeros(hi.size, dtype=np.uint64)
    n = lib().pcvo_get_visible_nodes(_d(_vec3(bmin)), _d(_vec3(bmax)), hi.size, hi.ctypes.data_as(_u64p),
                                     lo.ctypes.data_as(_u64p), npnts.ctypes.data_as(_i64p), _d(_f64(matrix)),
                                     ohi.ctypes.data_as(_u64p), olo.ctypes.data_as(_u64p))
    if n < 0:
        return None
    return [node_id_str(int(ohi[i]), int(olo[i])) for i in range(n)]


def nodes_in_location(bmin, bmax, nodes, kind, params):
    hi, lo, npnts = _tree_arrays(nodes)
    ohi, olo = np.zeros(hi.size, dtype=np.uint64), np.zeros(hi.size, dtype=np.uint64)
    pr = _f64(params if params is not None else [0.0])
    n = lib().pcvo_nodes_in_location(_d(_vec3(bmin)), _d(_vec3(bmax)), hi.size, hi.ctypes.data_as(_u64p),
                                     lo.ctypes.data_as(_u64p), npnts.ctypes.data_as(_i64p), kind, _d(pr),
                                     ohi.ctypes.data_as(_u64p), olo.ctypes.data_as(_u64p))
    if n < 0:
        return None
    return [node_id_str(int(ohi[i]), int(olo[i])) for i in range(n)]


def cull_points(kind, params, x, y, z, attr=None, interval=None):
    x, y, z = _f64(x), _f64(y), _f64(z)
    keep = np.zeros(x.size, dtype=np.uint8)
    ap = None
    if attr is not None:
        attr = np.ascontiguousarray(attr, dtype=np.float32)
        ap = attr.ctypes.data_as(_fp)
    iv = _f64(interval) if interval is not None else None
    lib().pcvo_cull_points(kind, _d(_f64(params if params is not None else [0.0])), x.size, _d(x), _d(y), _d(z), ap,
                           _d(iv) if iv is not None else None, keep.ctypes.data_as(_u8p))
    return keep


def iso_transform_points(iso7, x, y, z):
    x, y, z = _f64(x), _f64(y), _f64(z)
    ox, oy, oz = np.zeros_like(x), np.zeros_like(x), np.zeros_like(x)
    lib().pcvo_iso_transform_points(_d(_f64(iso7)), x.size, _d(x), _d(y), _d(z), _d(ox), _d(oy), _d(oz))
    return ox, oy, oz


def decode_positions(enc, cube_min, edge, xyz_bytes):
    bpc = {1: 1, 2: 2, 3: 4, 4: 8}[enc]
    buf = np.frombuffer(xyz_bytes, dtype=np.uint8).copy()
    n = buf.size // (3 * bpc)
    x, y, z = np.zeros(n), np.zeros(n), np.zeros(n)
    lib().pcvo_decode_positions(enc, _d(_vec3(cube_min)), float(edge), n, buf.ctypes.data_as(_u8p), _d(x), _d(y), _d(z))
    return x, y, z
