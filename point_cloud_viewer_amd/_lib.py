"""ctypes binding of libpcv_hip.so (the C ABI of include/pcv_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCV_HIP_LIBRARY: an alternative in-tree build of the same sources — "exp" = libpcv_hip_exp.so, the build with
# -DPCV_EXPERIMENTS whose environment switches are live (A/B scripts under tools/, tests of the alternative kernels), or
# a path (tools/build_variants.sh). The default library reads no environment variable.
_alt = os.environ.get("PCV_HIP_LIBRARY")
LIB_PATH = os.path.join(_HERE, "libpcv_hip_exp.so") if _alt == "exp" else (_alt or os.path.join(_HERE, "libpcv_hip.so"))

PCV_OK = 0
ABI_VERSION = 2  # include/pcv_hip.h PCV_ABI_VERSION
PCV_E_INVALID, PCV_E_HIP, PCV_E_IO, PCV_E_OOM, PCV_E_DEPTH, PCV_E_NOT_FOUND = -1, -2, -3, -4, -5, -6
MEM_HOST, MEM_DEVICE = 0, 1
ENC_UINT8, ENC_UINT16, ENC_FLOAT32, ENC_FLOAT64 = 1, 2, 3, 4
BUILD_COMPUTE_BBOX = 1
BUILD_NO_SPECULATION = 2
BUILD_FORCE_SINGLE_CHAIN = 4
BUILD_NO_SINGLE_CHAIN = 8
BUILD_CHECK_RESOLVE = 16
BUILD_STAGE_TIMES = 32
ROUTE_OCTANTS_ONLY = 64
MAX_KEY_LEVELS = 21
NUM_STAGES = 10
STAGE_NAMES = ["aabb", "chain_keys", "sort_keys", "node_split", "table", "leaf_encode", "sort_records",
               "promote_encode", "sort_second", "total"]

_ERR_NAMES = {PCV_E_INVALID: "PCV_E_INVALID", PCV_E_HIP: "PCV_E_HIP", PCV_E_IO: "PCV_E_IO", PCV_E_OOM: "PCV_E_OOM",
              PCV_E_DEPTH: "PCV_E_DEPTH", PCV_E_NOT_FOUND: "PCV_E_NOT_FOUND"}


class PcvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Points(C.Structure):
    _fields_ = [("n", C.c_uint64), ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("color", C.c_void_p),
                ("color_stride", C.c_uint32), ("intensity", C.c_void_p), ("mem", C.c_int32)]


class NodeCopy(C.Structure):
    _fields_ = [("node", C.c_uint64), ("dst_offset", C.c_uint64 * 3)]


class BuildParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("bbox_min", C.c_double * 3), ("bbox_max", C.c_double * 3),
                ("max_points_per_node", C.c_uint32), ("flags", C.c_uint32)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("params", C.c_double * 32)]


SHAPE_ALL, SHAPE_AABB, SHAPE_FRUSTUM, SHAPE_OBB, SHAPE_FRUSTUM_WITH_INVERSE = 0, 1, 2, 3, 4
REL_IN, REL_CROSS, REL_OUT = 0, 1, 2


class RouteState(C.Structure):
    _fields_ = [("cx", C.c_void_p), ("cy", C.c_void_p), ("cz", C.c_void_p), ("oct_rgb", C.c_void_p)]


class RouteDst(C.Structure):
    _fields_ = [("oct_rgb", C.c_void_p), ("cx", C.c_void_p), ("cy", C.c_void_p), ("cz", C.c_void_p), ("intensity", C.c_void_p)]


class Plane(C.Structure):
    _fields_ = [("src", C.c_void_p), ("elem_bytes", C.c_uint32)]


class RoutedPoints(C.Structure):
    _fields_ = [("n", C.c_uint64), ("cx", C.c_void_p), ("cy", C.c_void_p), ("cz", C.c_void_p), ("oct_rgb", C.c_void_p),
                ("intensity", C.c_void_p)]


class TopStreams(C.Structure):
    _fields_ = [("l1", C.c_uint64 * 8), ("l2", C.c_uint64 * 64), ("l1_split_mask", C.c_uint32), ("reserved", C.c_uint32)]


class TopLayout(C.Structure):
    _fields_ = [("root_points", C.c_uint64), ("l1_stream", C.c_uint64 * 8), ("l1_offset", C.c_uint32 * 8),
                ("l2_offset", C.c_uint32 * 64)]


class SplitNode(C.Structure):
    _fields_ = [("id_high", C.c_uint64), ("id_low", C.c_uint64), ("first", C.c_uint64), ("count", C.c_uint64),
                ("level", C.c_uint32), ("parent", C.c_uint32), ("first_child", C.c_uint32), ("child_mask", C.c_uint32),
                ("is_leaf", C.c_uint32), ("reserved", C.c_uint32)]


class PromoteNode(C.Structure):
    _fields_ = [("stream_len", C.c_uint64), ("num_points", C.c_uint64), ("child_offset", C.c_uint64)]


class NodeInfo(C.Structure):
    _fields_ = [("id_high", C.c_uint64), ("id_low", C.c_uint64), ("num_points", C.c_int64), ("level", C.c_uint32),
                ("encoding", C.c_uint32), ("cube_min", C.c_double * 3), ("cube_edge", C.c_double),
                ("xyz_offset", C.c_uint64), ("point_offset", C.c_uint64)]


# every symbol include/pcv_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_SIGNATURES = {
    "pcv_abi_version": (C.c_int, []),
    "pcv_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "pcv_ctx_destroy": (None, [_vp]),
    "pcv_last_error": (C.c_char_p, [_vp]),
    "pcv_ctx_trim": (C.c_int, [_vp]),
    "pcv_ctx_synchronize": (C.c_int, [_vp]),
    "pcv_ctx_wait_stream": (C.c_int, [_vp, _vp]),
    "pcv_ctx_signal_stream": (C.c_int, [_vp, _vp]),
    "pcv_ctx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "pcv_ctx_reset_kernel_stats": (C.c_int, [_vp]),
    "pcv_ctx_kernel_stats": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_double)]),
    "pcv_build_octree": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), C.POINTER(_vp)]),
    "pcv_ingest_begin": (C.c_int, [_vp, C.c_uint64, C.c_int, C.POINTER(_vp)]),
    "pcv_ingest_append": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64]),
    "pcv_ingest_num_points": (C.c_uint64, [_vp]),
    "pcv_ingest_bbox": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pcv_ingest_finish": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(_vp)]),
    "pcv_ingest_abort": (None, [_vp]),
    "pcv_octree_num_nodes": (C.c_uint64, [_vp]),
    "pcv_octree_num_points": (C.c_uint64, [_vp]),
    "pcv_octree_has_intensity": (C.c_int, [_vp]),
    "pcv_octree_node": (C.c_int, [_vp, C.c_uint64, C.POINTER(NodeInfo)]),
    "pcv_octree_meta": (None, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_int)]),
    "pcv_octree_node_data": (C.c_int, [_vp, C.c_uint64, C.c_int, C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "pcv_octree_device_blob": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "pcv_octree_write_dir": (C.c_int, [_vp, C.c_char_p]),
    "pcv_octree_free": (None, [_vp]),
    "pcv_octree_stage_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int]),
    "pcv_octree_build_info": (None, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pcv_octree_spec_stats": (None, [_vp, C.POINTER(C.c_uint64)]),
    "pcv_octree_record_bytes": (C.c_int, [_vp]),
    "pcv_octree_spec_continued": (C.c_uint64, [_vp]),
    "pcv_octree_wide_pool_entries": (C.c_uint64, [_vp]),
    "pcv_octree_settled_in_sort": (C.c_uint64, [_vp]),
    "pcv_aabb_reduce": (C.c_int, [_vp, C.POINTER(Points), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pcv_level_table": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int,
                                  C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "pcv_level_shortcuts": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_double)]),
    "pcv_chain_keys": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), C.c_int, _vp]),
    "pcv_route_buckets": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), _vp, C.POINTER(C.c_uint64),
                                    C.POINTER(RouteState)]),
    "pcv_route_tiles": (C.c_uint64, [C.c_uint64]),
    "pcv_route_plan": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), _vp, _vp, C.POINTER(C.c_uint64)]),
    "pcv_route_scatter": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), _vp, _vp, C.c_uint32, _vp, C.POINTER(RouteDst)]),
    "pcv_partition_by_owner": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint32, _vp, C.c_uint32, C.POINTER(Plane),
                                         C.POINTER(C.c_void_p)]),
    "pcv_build_begin_routed": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(RoutedPoints), C.POINTER(_vp)]),
    "pcv_octree_write_nodes": (C.c_int, [_vp, C.c_char_p, C.c_uint32]),
    "pcv_write_meta": (C.c_int, [C.c_char_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(NodeInfo),
                                 C.c_uint64]),
    "pcv_octree_copy_node": (C.c_int, [_vp, C.c_uint64, C.c_int, _vp, C.c_uint64, C.c_int]),
    "pcv_octree_copy_nodes": (C.c_int, [_vp, C.POINTER(NodeCopy), C.c_uint64, _vp, C.c_uint64, C.c_int]),
    "pcv_build_begin": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), C.POINTER(_vp)]),
    "pcv_build_top_streams": (C.c_int, [_vp, C.POINTER(TopStreams)]),
    "pcv_build_finish": (C.c_int, [_vp, C.POINTER(TopLayout)]),
    "pcv_node_split": (C.c_int, [_vp, C.POINTER(BuildParams), _vp, C.c_uint64, C.c_int, C.POINTER(SplitNode), C.c_uint64,
                                 C.POINTER(C.c_uint64)]),
    "pcv_promote_assign": (C.c_int, [C.POINTER(SplitNode), C.c_uint64, C.POINTER(PromoteNode), C.c_uint64, _vp, _vp]),
    "pcv_gather_encode": (C.c_int, [_vp, C.POINTER(BuildParams), C.POINTER(Points), C.POINTER(SplitNode), C.c_uint64,
                                    C.POINTER(_vp)]),
    "pcv_sort_keys64": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int]),
    "pcv_sort_keys32": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int]),
    "pcv_sort_pairs32": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int]),
    "pcv_selftest_division": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pcv_ply_read": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.c_char_p, C.c_uint64]),
    "pcv_ply_num_points": (C.c_uint64, [_vp]),
    "pcv_ply_points": (C.c_int, [_vp, C.POINTER(Points)]),
    "pcv_ply_free": (None, [_vp]),
    "pcv_build_octree_from_ply": (C.c_int, [_vp, C.POINTER(BuildParams), C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "pcv_octree_open_dir": (C.c_int, [_vp, C.c_char_p, C.POINTER(_vp)]),
    "pcv_shapes_create": (C.c_int, [_vp, C.POINTER(Shape), C.c_uint32, C.POINTER(_vp)]),
    "pcv_shapes_free": (None, [_vp]),
    "pcv_shapes_count": (C.c_uint32, [_vp]),
    "pcv_shapes_get": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "pcv_cull_nodes": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "pcv_cull_nodes_sparse": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp]),
    "pcv_visible_nodes": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp]),
    "pcv_nodes_in_location": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    "pcv_cull_points": (C.c_int, [_vp, _vp, C.c_uint32, C.POINTER(Points), C.POINTER(C.c_double), _vp,
                                  C.POINTER(C.c_uint64)]),
    "pcv_cull_node_points": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_double), _vp,
                                       C.POINTER(C.c_uint64)]),
    "pcv_query_points": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.POINTER(C.c_double), C.c_uint64, C.c_int, _vp, _vp, _vp, _vp,
                                   _vp, C.POINTER(C.c_uint64)]),
    "pcv_query_node_points": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_double), C.c_uint64, C.c_int, _vp, _vp,
                                        _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "pcv_octree_nodes_blob": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pcv_transform_points": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(Points), _vp, _vp, _vp]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load_library():
    """Load libpcv_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C point_cloud_viewer_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pcv_abi_version() != ABI_VERSION:  # a stale libpcv_hip.so next to newer bindings: fail loudly, not subtly
        raise ImportError(f"{LIB_PATH} has ABI version {lib.pcv_abi_version()}, these bindings are written for {ABI_VERSION}: "
                          "rebuild it (make -C point_cloud_viewer_amd/csrc)")
    _lib = lib
    return lib
