// pcv_ply_layout.h — what pcv_ply_parse_header (pcv_ply.cpp) learns from the header of a binary little-endian PLY: enough
// to decode the vertex records, on the host (pcv_ply_read) or on the device (pcv_build_octree_from_ply).
#pragma once
#include <stdint.h>
#include <stdio.h>

enum PcvPlyType { PCV_PLY_I8, PCV_PLY_U8, PCV_PLY_I16, PCV_PLY_U16, PCV_PLY_I32, PCV_PLY_U32, PCV_PLY_F32, PCV_PLY_F64 };

struct PcvPlyLayout {
  long long vertex_count = 0;
  int stride = 0;        // bytes per vertex record
  long body_offset = 0;  // file offset of the first record
  int x_type = 0, x_off = 0, y_type = 0, y_off = 0, z_type = 0, z_off = 0;  // PcvPlyType + byte offset inside the record
  int r_off = -1, g_off = -1, b_off = -1;  // uchar colour properties, -1: none
  int i_off = -1;                          // float `intensity`, -1: none
  double offset[3] = {0, 0, 0};            // `comment offset: x y z`
};
// Leaves `f` at the first vertex record; rejects vertex counts the rest of the file cannot hold.
int pcv_ply_parse_header(FILE* f, PcvPlyLayout* lay, char* err, uint64_t errcap);
