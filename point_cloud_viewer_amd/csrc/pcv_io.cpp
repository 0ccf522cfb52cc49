// pcv_io.cpp — writes a finished octree in the reference's on-disk layout (host side, no GPU work).
//
//   <NodeId>.xyz        n * 3 * {1,2,4,8} bytes, little endian, AoS xyz   reference src/read_write/raw.rs:374-392
//   <NodeId>.rgb        n * 3 bytes                                        src/lib.rs:74-80 (attribute_extension)
//   <NodeId>.intensity  n * 4 bytes LE f32
//   files of a node with zero points do not exist                          src/read_write/node_writer.rs:78-89
//   meta.pb             proto3 `Meta`, version 13                          point_viewer_proto_rust/src/proto.proto:136-149
// NodeId Display: "r" + index in octal, zero-padded to `level` digits      src/octree/node.rs:73-86
#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pcv_internal.h"

namespace {

std::string node_name(const pcv_node_info& n) {
  std::string s = "r";
  unsigned __int128 index = ((unsigned __int128)(n.id_high & 0x00ffffffffffffffull) << 64) | n.id_low;
  for (int j = (int)n.level - 1; j >= 0; --j) s.push_back((char)('0' + (int)((index >> (3 * j)) & 7)));
  return s;
}

void varint(std::vector<uint8_t>& o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back((uint8_t)(v | 0x80));
    v >>= 7;
  }
  o.push_back((uint8_t)v);
}
void tag(std::vector<uint8_t>& o, int field, int wire) { varint(o, ((uint64_t)field << 3) | (uint64_t)wire); }
void f64_field(std::vector<uint8_t>& o, int field, double d) {
  if (d == 0.) return;  // proto3: default values are not serialised
  tag(o, field, 1);
  uint64_t u;
  std::memcpy(&u, &d, 8);
  for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(u >> (8 * i)));
}
void bytes_field(std::vector<uint8_t>& o, int field, const std::vector<uint8_t>& b) {
  tag(o, field, 2);
  varint(o, b.size());
  o.insert(o.end(), b.begin(), b.end());
}
std::vector<uint8_t> vec3d(const double v[3]) {  // proto.proto:33-37 Vector3d
  std::vector<uint8_t> o;
  f64_field(o, 1, v[0]);
  f64_field(o, 2, v[1]);
  f64_field(o, 3, v[2]);
  return o;
}

bool write_file(const std::string& path, const uint8_t* data, uint64_t len) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = len == 0 || fwrite(data, 1, len, f) == len;
  return fclose(f) == 0 && ok;
}

}  // namespace

// octree/mod.rs:87-99 to_meta_proto + node.rs:260-270 to_node_proto + node.rs:101-106 NodeId::to_proto
std::vector<uint8_t> pcv_encode_meta(const pcv_octree* t) {
  std::vector<uint8_t> octree;
  f64_field(octree, 2, t->resolution);
  for (const pcv_node_info& n : t->nodes) {
    std::vector<uint8_t> node, id;
    tag(node, 2, 0);
    varint(node, n.encoding);
    if (n.num_points != 0) {
      tag(node, 3, 0);
      varint(node, (uint64_t)n.num_points);
    }
    if (n.id_high != 0) {
      tag(id, 3, 0);
      varint(id, n.id_high);
    }
    if (n.id_low != 0) {
      tag(id, 4, 0);
      varint(id, n.id_low);
    }
    bytes_field(node, 4, id);  // the id sub-message is always present (octree/mod.rs:199 unwraps it)
    bytes_field(octree, 3, node);
  }
  std::vector<uint8_t> cuboid;
  bytes_field(cuboid, 3, vec3d(t->bbox_min));
  bytes_field(cuboid, 4, vec3d(t->bbox_max));
  std::vector<uint8_t> meta;
  tag(meta, 1, 0);
  varint(meta, 13);  // CURRENT_VERSION src/lib.rs:48
  bytes_field(meta, 4, cuboid);
  bytes_field(meta, 6, octree);
  return meta;
}

extern "C" int pcv_octree_write_dir(pcv_octree* t, const char* directory) {
  if (!t || !directory) return PCV_E_INVALID;
  pcv_ctx* ctx = t->ctx;
  int rc = pcv_octree_fetch_host(t);
  if (rc) return rc;
  std::string dir(directory);
  ::mkdir(dir.c_str(), 0777);  // generation.rs:308 "Ignore errors, maybe directory is already there."
  struct stat st;
  if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return ctx->fail(PCV_E_IO, "cannot create directory " + dir);
  for (const pcv_node_info& n : t->nodes) {
    if (n.num_points == 0) continue;
    const std::string stem = dir + "/" + node_name(n);
    const uint64_t np = (uint64_t)n.num_points;
    if (!write_file(stem + ".xyz", t->h_xyz.data() + n.xyz_offset, np * 3 * (uint64_t)pcv_bytes_per_coordinate(n.encoding)))
      return ctx->fail(PCV_E_IO, "cannot write " + stem + ".xyz");
    if (!write_file(stem + ".rgb", t->h_rgb.data() + n.point_offset * 3, np * 3))
      return ctx->fail(PCV_E_IO, "cannot write " + stem + ".rgb");
    if (t->has_intensity && !write_file(stem + ".intensity", t->h_int.data() + n.point_offset * 4, np * 4))
      return ctx->fail(PCV_E_IO, "cannot write " + stem + ".intensity");
  }
  std::vector<uint8_t> meta = pcv_encode_meta(t);
  if (!write_file(dir + "/meta.pb", meta.data(), meta.size())) return ctx->fail(PCV_E_IO, "cannot write meta.pb");
  return PCV_OK;
}
