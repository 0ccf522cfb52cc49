// pcv_oracle_core.h — CPU ORACLE (test infrastructure only, never shipped, never on the product path).
//
// Plain C++17 restatement of the arithmetic of point_cloud_viewer's octree build path.
// Every function cites the reference file:line it follows (paths relative to /root/reference).
//
// Parity status: the reference is Rust and cannot be compiled in this environment (no cargo/rustc),
// so this restatement is pinned against the reference's own known-answer tests
// (codec.rs:154-212, node.rs:277-317, octree/tests.rs:18-46, sat.rs:214-268, obb.rs:100-141,
// math/mod.rs:191-220) in tests/test_oracle_kats.py — NOT against a run of the Rust binary.
// Third-party semantics taken as rules (unpinned by reference tests): simba 0.2.1 `try_convert`
// f64->u8/u16/f32 == Rust `as` casts (truncate toward zero, saturate, NaN->0 / RNE for f32);
// num 0.3 `clamp` == if/else-if chain (NaN passes through).
//
// Build flags that matter: -ffp-contract=off (no implicit FMA), no -ffast-math. FMA appears only
// where the reference uses f64::mul_add (codec.rs:130,138) and is written as std::fma.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace pcvo {

typedef unsigned __int128 u128;

// proto.proto:82-88 PositionEncoding enum values.
enum Enc : uint8_t { ENC_INVALID = 0, ENC_U8 = 1, ENC_U16 = 2, ENC_F32 = 3, ENC_F64 = 4 };

// codec.rs:64-71 bytes_per_coordinate
static inline int bytes_per_coordinate(Enc e) {
  switch (e) {
    case ENC_U8: return 1;
    case ENC_U16: return 2;
    case ENC_F32: return 4;
    default: return 8;
  }
}

struct Aabb {  // aabb.rs:13-16
  double mn[3];
  double mx[3];
};

struct Cube {  // aabb.rs:143-147
  double mn[3];
  double edge;
};

// aabb.rs:149-157 Cube::bounding — f64::max chain over the three extents.
static inline Cube cube_bounding(const Aabb& b) {
  Cube c;
  double e = std::fmax(std::fmax(b.mx[0] - b.mn[0], b.mx[1] - b.mn[1]), b.mx[2] - b.mn[2]);
  c.mn[0] = b.mn[0];
  c.mn[1] = b.mn[1];
  c.mn[2] = b.mn[2];
  c.edge = e;
  return c;
}

// aabb.rs:175-192 Cube::max then Cube::center: ((min) + (min + edge)) / 2 per component.
static inline void cube_center(const Cube& c, double out[3]) {
  for (int a = 0; a < 3; ++a) {
    double mx = c.mn[a] + c.edge;
    out[a] = (c.mn[a] + mx) / 2.;
  }
}

// node.rs:34-42 ChildIndex::from_bounding_cube — strict '>' against the centre; x=bit2,y=bit1,z=bit0.
static inline uint8_t child_index_from_bounding_cube(const Cube& c, const double p[3]) {
  double ctr[3];
  cube_center(c, ctr);
  uint8_t gx = p[0] > ctr[0];
  uint8_t gy = p[1] > ctr[1];
  uint8_t gz = p[2] > ctr[2];
  return (uint8_t)((gx << 2) | (gy << 1) | gz);
}

// node.rs:56-173 NodeId: u128, level in the top 8 bits, index in the low 120.
struct NodeId {
  u128 v;
  static NodeId from_level_index(uint8_t level, u128 index) {  // node.rs:108-111
    NodeId n;
    n.v = ((u128)level << 120) | index;
    return n;
  }
  static NodeId root() { return NodeId{0}; }                        // node.rs:114-116
  uint8_t level() const { return (uint8_t)(v >> 120); }            // node.rs:147-149
  u128 index() const { return v & ((((u128)1) << 120) - 1); }      // node.rs:152-154
  NodeId get_child_id(uint8_t child) const {                        // node.rs:120-125
    return from_level_index((uint8_t)(level() + 1), (index() << 3) + (u128)child);
  }
  bool parent_id(NodeId* out) const {                               // node.rs:136-144
    if (level() == 0) return false;
    *out = from_level_index((uint8_t)(level() - 1), index() >> 3);
    return true;
  }
  uint8_t child_index() const { return (uint8_t)(index() & 7); }   // node.rs:128-133
  uint64_t high() const { return (uint64_t)(v >> 64); }            // node.rs:101-106
  uint64_t low() const { return (uint64_t)v; }
  static NodeId from_high_low(uint64_t h, uint64_t l) {             // node.rs:89-99 (non-deprecated arm)
    return NodeId{((u128)h << 64) | (u128)l};
  }
  // node.rs:73-86 Display: "r" + index in octal, zero padded to `level` digits.
  std::string to_string() const {
    std::string s = "r";
    int lv = level();
    u128 idx = index();
    for (int j = lv - 1; j >= 0; --j) s.push_back((char)('0' + (int)((idx >> (3 * j)) & 7)));
    return s;
  }
  // node.rs:59-70 FromStr
  static NodeId from_string(const std::string& s) {
    uint8_t level = (uint8_t)(s.size() - 1);
    u128 idx = 0;
    for (size_t i = 1; i < s.size(); ++i) idx = (idx << 3) | (u128)(s[i] - '0');
    return from_level_index(level, idx);
  }
  // node.rs:157-172 find_bounding_cube: iterative, most significant digit first,
  // edge /= 2 then min += bit * edge.
  Cube find_bounding_cube(const Cube& root) const {
    double edge = root.edge;
    double mn[3] = {root.mn[0], root.mn[1], root.mn[2]};
    for (int level = (int)this->level() - 1; level >= 0; --level) {
      edge /= 2.;
      unsigned ci = (unsigned)((v >> (3 * level)) & 7);
      unsigned z = ci & 1, y = (ci >> 1) & 1, x = (ci >> 2) & 1;
      mn[0] += (double)x * edge;
      mn[1] += (double)y * edge;
      mn[2] += (double)z * edge;
    }
    Cube c;
    c.mn[0] = mn[0];
    c.mn[1] = mn[1];
    c.mn[2] = mn[2];
    c.edge = edge;
    return c;
  }
  bool operator==(const NodeId& o) const { return v == o.v; }
  bool operator<(const NodeId& o) const { return v < o.v; }
};

// Rust `f64 as u32`: truncate toward zero, saturate, NaN -> 0.
static inline uint32_t rust_f64_as_u32(double x) {
  if (!(x > 0.0)) return 0;  // NaN, negatives, zero
  if (x >= 4294967295.0) return 4294967295u;
  return (uint32_t)x;
}

// codec.rs:31-40 PositionEncoding::new. `+ 1` wraps in release builds.
static inline Enc position_encoding(double edge_length, double resolution) {
  uint32_t min_bits = rust_f64_as_u32(std::log2(edge_length / resolution)) + 1u;
  if (min_bits <= 8) return ENC_U8;
  if (min_bits <= 16) return ENC_U16;
  if (min_bits <= 24) return ENC_F32;
  return ENC_F64;
}

// num 0.3 clamp, as used at codec.rs:89,97,142-148 (NaN falls through unchanged).
static inline double clamp01(double v) {
  if (v < 0.) return 0.;
  else if (v > 1.) return 1.;
  else return v;
}

// codec.rs:102-113 vec3_fixpoint_encode (per component), simba try_convert == `as` cast.
static inline uint64_t fixpoint_encode(double value, double mn, double edge, double maxval) {
  double t = clamp01((value - mn) / edge);
  double s = maxval * t;
  if (!(s > 0.0)) return 0;  // NaN -> 0, negative/zero -> 0
  if (s >= maxval) return (uint64_t)maxval;
  return (uint64_t)s;  // truncation toward zero
}

// Raw code of one coordinate under encoding `e`: the integer value (u8/u16) or the IEEE bit pattern
// (f32/f64) that goes to disk little-endian. codec.rs:102-121.
static inline uint64_t encode_coord(Enc e, double value, double mn, double edge) {
  switch (e) {
    case ENC_U8: return fixpoint_encode(value, mn, edge, 255.0);
    case ENC_U16: return fixpoint_encode(value, mn, edge, 65535.0);
    case ENC_F32: {
      float f = (float)clamp01((value - mn) / edge);  // RNE
      uint32_t u;
      std::memcpy(&u, &f, 4);
      return u;
    }
    default: {
      double d = clamp01((value - mn) / edge);
      uint64_t u;
      std::memcpy(&u, &d, 8);
      return u;
    }
  }
}

// codec.rs:124-139 fixpoint_decode / decode: (v / max).mul_add(edge, min) — single-rounding FMA.
static inline double decode_coord(Enc e, uint64_t raw, double mn, double edge) {
  switch (e) {
    case ENC_U8: return std::fma((double)raw / 255.0, edge, mn);
    case ENC_U16: return std::fma((double)raw / 65535.0, edge, mn);
    case ENC_F32: {
      uint32_t u = (uint32_t)raw;
      float f;
      std::memcpy(&f, &u, 4);
      return std::fma((double)f, edge, mn);
    }
    default: {
      double d;
      std::memcpy(&d, &raw, 8);
      return std::fma(d, edge, mn);
    }
  }
}

static inline void put_le(uint8_t* dst, uint64_t raw, int nbytes) {
  for (int i = 0; i < nbytes; ++i) dst[i] = (uint8_t)(raw >> (8 * i));
}
static inline uint64_t get_le(const uint8_t* src, int nbytes) {
  uint64_t r = 0;
  for (int i = 0; i < nbytes; ++i) r |= (uint64_t)src[i] << (8 * i);
  return r;
}

// generation.rs:37 — a compile-time constant in the reference. The oracle keeps it in a variable ONLY so
// tests can build deep trees from small clouds (pcvo_set_max_points_per_node); default = reference.
extern int64_t MAX_POINTS_PER_NODE;
// lib.rs:52
static const size_t NUM_POINTS_PER_BATCH = 500000;
// lib.rs:48
static const int CURRENT_VERSION = 13;

}  // namespace pcvo
