"""MI355X-native octree build / cull hot path of point_cloud_viewer (hand-written HIP for gfx950 behind a C ABI).

The package is a thin host mirror of the reference's interface for this path; all compute lives in
libpcv_hip.so (point_cloud_viewer_amd/csrc). Importing does not require a GPU; calling does.
"""
from . import _lib
from ._lib import (PCV_E_DEPTH, PCV_E_HIP, PCV_E_INVALID, PCV_E_IO, PCV_E_NOT_FOUND, PCV_E_OOM, PCV_OK,  # noqa: F401
                   PcvError, load_library)
from .octree import (Aabb, Context, OctreeResult, Shapes, build_octree, build_octree_from_file, level_shortcuts,  # noqa: F401
                     level_table,
                     node_name, read_ply)

__all__ = ["Aabb", "Context", "OctreeResult", "build_octree", "level_table", "node_name", "PcvError",
           "load_library"]
