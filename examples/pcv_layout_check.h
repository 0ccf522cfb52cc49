/* Struct layouts the FFI mirrors (Rust #[repr(C)] in point_cloud_viewer_amd/rust_shim/src/lib.rs, ctypes in
 * point_cloud_viewer_amd/_lib.py) rely on, asserted by the C and the C++ compiler against include/pcv_hip.h.
 * LP64, natural alignment — what #[repr(C)] lays out on x86_64-unknown-linux-gnu. */
#ifndef PCV_LAYOUT_CHECK_H
#define PCV_LAYOUT_CHECK_H
#include <stddef.h>

#include "pcv_hip.h"

#ifdef __cplusplus
#define PCV_SA(cond, msg) static_assert(cond, msg)
#else
#define PCV_SA(cond, msg) _Static_assert(cond, msg)
#endif

PCV_SA(sizeof(pcv_points) == 64, "pcv_points");
PCV_SA(offsetof(pcv_points, n) == 0 && offsetof(pcv_points, x) == 8 && offsetof(pcv_points, y) == 16 &&
           offsetof(pcv_points, z) == 24 && offsetof(pcv_points, color) == 32 && offsetof(pcv_points, color_stride) == 40 &&
           offsetof(pcv_points, intensity) == 48 && offsetof(pcv_points, mem) == 56,
       "pcv_points fields");
PCV_SA(sizeof(pcv_build_params) == 64, "pcv_build_params");
PCV_SA(offsetof(pcv_build_params, resolution) == 0 && offsetof(pcv_build_params, bbox_min) == 8 &&
           offsetof(pcv_build_params, bbox_max) == 32 && offsetof(pcv_build_params, max_points_per_node) == 56 &&
           offsetof(pcv_build_params, flags) == 60,
       "pcv_build_params fields");
PCV_SA(sizeof(pcv_node_info) == 80, "pcv_node_info");
PCV_SA(offsetof(pcv_node_info, id_high) == 0 && offsetof(pcv_node_info, id_low) == 8 && offsetof(pcv_node_info, num_points) == 16 &&
           offsetof(pcv_node_info, level) == 24 && offsetof(pcv_node_info, encoding) == 28 && offsetof(pcv_node_info, cube_min) == 32 &&
           offsetof(pcv_node_info, cube_edge) == 56 && offsetof(pcv_node_info, xyz_offset) == 64 &&
           offsetof(pcv_node_info, point_offset) == 72,
       "pcv_node_info fields");
PCV_SA(sizeof(pcv_shape) == 264 && offsetof(pcv_shape, kind) == 0 && offsetof(pcv_shape, params) == 8, "pcv_shape");
PCV_SA(sizeof(pcv_top_streams) == 584 && offsetof(pcv_top_streams, l2) == 64 && offsetof(pcv_top_streams, l1_split_mask) == 576,
       "pcv_top_streams");
PCV_SA(sizeof(pcv_top_layout) == 360 && offsetof(pcv_top_layout, l1_stream) == 8 && offsetof(pcv_top_layout, l1_offset) == 72 &&
           offsetof(pcv_top_layout, l2_offset) == 104,
       "pcv_top_layout");
PCV_SA(sizeof(pcv_routed_points) == 48 && offsetof(pcv_routed_points, oct_rgb) == 32 && offsetof(pcv_routed_points, intensity) == 40,
       "pcv_routed_points");
PCV_SA(sizeof(pcv_route_state) == 32, "pcv_route_state");
PCV_SA(sizeof(pcv_plane) == 16 && offsetof(pcv_plane, elem_bytes) == 8, "pcv_plane");
PCV_SA(sizeof(pcv_split_node) == 56 && offsetof(pcv_split_node, first) == 16 && offsetof(pcv_split_node, level) == 32 &&
           offsetof(pcv_split_node, is_leaf) == 48,
       "pcv_split_node");
PCV_SA(sizeof(pcv_promote_node) == 24 && offsetof(pcv_promote_node, child_offset) == 16, "pcv_promote_node");

#endif
