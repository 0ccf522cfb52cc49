// pcv_query.hip — batched frustum / OBB / AABB transform-and-cull for gfx950.
//
//   K7a shape_setup      Frustum::from_matrix4 / intersector / cache_separating_axes_for_aabb
//                        (reference src/geometry/frustum.rs:111-166, src/math/sat.rs:111-143), Obb (obb.rs:48-80)
//   K7  cull_nodes       sat() of every (shape, node cube) pair (sat.rs:174-205) + relative_size_on_screen
//                        (src/octree/mod.rs:119-139)
//   K7b visible_nodes    Octree::get_visible_nodes — best-first traversal with Rust's BinaryHeap order
//                        (octree/mod.rs:228-283,360-404), one lane per frustum
//   K7c nodes_in_location  NodeIdsIterator BFS (src/octree/octree_iterator.rs, octree/mod.rs:309-323)
//   K8  cull_points      FilteredIterator keep mask (src/iterator.rs:96-119; frustum.rs:120-125, obb.rs:83-90,
//                        aabb.rs:46-48), on raw f64 positions or on a node's encoded bytes decoded on the fly
//                        (src/read_write/codec.rs:124-139)
//   K9  transform_points Isometry3 * Point3 (xray/src/generation.rs:493-497)
//
// Arithmetic follows the nalgebra 0.22 formulas restated in DESIGN.md ("query arithmetic"): left-to-right dot
// products, gemv column accumulation, division by the norm, no fused multiply-add (-ffp-contract=off).
// Bounds: K7 is f64-VALU bound (about 1 kflop per pair on 128 B of data), K8/K9 are HBM streams.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "pcv_chain_dev.h"

// ---------------------------------------------------------------------------------------------
// device math
// ---------------------------------------------------------------------------------------------
struct V3d {
  double x, y, z;
};
__host__ __device__ __forceinline__ V3d v_sub(V3d a, V3d b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ __forceinline__ V3d v_add(V3d a, V3d b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ __forceinline__ double v_dot(V3d a, V3d b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__host__ __device__ __forceinline__ V3d v_cross(V3d a, V3d b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__host__ __device__ __forceinline__ V3d v_scale(V3d a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3d v_normalize(V3d v) {
  double n = sqrt(v_dot(v, v));
  return {v.x / n, v.y / n, v.z / n};
}
#define M4(m, r, c) (m)[(c) * 4 + (r)]

__device__ bool m4_try_inverse(const double* m, double* out) {
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.0) return false;
  double inv_det = 1.0 / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * inv_det;
  return true;
}

// nalgebra Matrix4::transform_point
__device__ __forceinline__ V3d m4_transform_point(const double* m, V3d p) {
  double r0 = ((M4(m, 0, 0) * p.x + M4(m, 0, 1) * p.y) + M4(m, 0, 2) * p.z) + M4(m, 0, 3);
  double r1 = ((M4(m, 1, 0) * p.x + M4(m, 1, 1) * p.y) + M4(m, 1, 2) * p.z) + M4(m, 1, 3);
  double r2 = ((M4(m, 2, 0) * p.x + M4(m, 2, 1) * p.y) + M4(m, 2, 2) * p.z) + M4(m, 2, 3);
  double n = ((M4(m, 3, 0) * p.x + M4(m, 3, 1) * p.y) + M4(m, 3, 2) * p.z) + M4(m, 3, 3);
  if (n != 0.0) return {r0 / n, r1 / n, r2 / n};
  return {r0, r1, r2};
}

__device__ __forceinline__ V3d quat_rotate(const double* q, V3d v) {  // UnitQuaternion * Vector3
  V3d qv = {q[0], q[1], q[2]};
  V3d t = v_scale(v_cross(qv, v), 2.0);
  V3d c = v_cross(qv, t);
  return v_add(v_add(v_scale(t, q[3]), c), v);
}

// ---------------------------------------------------------------------------------------------
// prepared shapes
// ---------------------------------------------------------------------------------------------
#define PCV_MAX_AXES 26
struct PcvShapeDev {
  int32_t kind;   // PCV_SHAPE_*
  int32_t valid;  // 0: matrix not invertible (Frustum::from_matrix4 -> None)
  int32_t naxes;
  int32_t pad;
  double clip_from_query[16];
  double query_from_clip[16];
  double iso[7];   // obb_from_query (translation xyz, quaternion ijkw) for contains()
  double half[3];
  double bmin[3], bmax[3];
  double corners[24];
  double axes[PCV_MAX_AXES * 3];
  double amin[PCV_MAX_AXES];  // projection interval of the shape's own corners on each axis
  double amax[PCV_MAX_AXES];
};

struct pcv_shapes {
  pcv_ctx* ctx;
  uint32_t count;
  PcvShapeDev* dev;
};

namespace {

__device__ void project8(const double* corners, V3d axis, double* mn, double* mx) {  // sat.rs:196-205
  double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
  for (int i = 0; i < 8; ++i) {
    double p = v_dot(V3d{corners[3 * i], corners[3 * i + 1], corners[3 * i + 2]}, axis);
    lo = fmin(lo, p);
    hi = fmax(hi, p);
  }
  *mn = lo;
  *mx = hi;
}

// cache_separating_axes against the unit edges / normals of an AABB (sat.rs:111-143)
__device__ void cache_axes_for_aabb(PcvShapeDev* s, const V3d* edges, int ne, const V3d* normals, int nn) {
  const V3d unit[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  V3d all[6 + 3 + 36];
  int na = 0;
  for (int i = 0; i < nn; ++i) all[na++] = normals[i];
  for (int i = 0; i < 3; ++i) all[na++] = unit[i];
  for (int i = 0; i < ne; ++i)
    for (int j = 0; j < 3; ++j) {
      V3d c = v_normalize(v_cross(edges[i], unit[j]));
      if (isfinite(c.x) && isfinite(c.y) && isfinite(c.z)) all[na++] = c;
    }
  int nd = 0;
  for (int i = 0; i < na; ++i) {
    bool dupe = false;
    for (int j = 0; j < nd; ++j) {
      V3d a2 = {s->axes[3 * j], s->axes[3 * j + 1], s->axes[3 * j + 2]};
      V3d dm = v_sub(all[i], a2), dp = v_add(all[i], a2);
      double d1 = v_dot(dm, dm), d2 = v_dot(dp, dp);
      if (fmin(d1, d2) < 2.220446049250313e-16) {
        dupe = true;
        break;
      }
    }
    if (!dupe && nd < PCV_MAX_AXES) {
      s->axes[3 * nd] = all[i].x;
      s->axes[3 * nd + 1] = all[i].y;
      s->axes[3 * nd + 2] = all[i].z;
      ++nd;
    }
  }
  s->naxes = nd;
}

__global__ __launch_bounds__(64) void shape_setup_kernel(PcvShapeDev* shapes, uint32_t count) {
  uint32_t f = blockIdx.x * 64 + threadIdx.x;
  if (f >= count) return;
  PcvShapeDev* s = shapes + f;
  s->valid = 1;
  if (s->kind == PCV_SHAPE_FRUSTUM || s->kind == PCV_SHAPE_FRUSTUM_WITH_INVERSE) {
    if (s->kind == PCV_SHAPE_FRUSTUM) {
      double inv[16];
      if (!m4_try_inverse(s->clip_from_query, inv)) {
        s->valid = 0;
        s->naxes = 0;
        return;
      }
      for (int i = 0; i < 16; ++i) s->query_from_clip[i] = inv[i];
    }
    const double sg[2] = {-1.0, 1.0};
    V3d k[8];
    int c = 0;
    for (int ix = 0; ix < 2; ++ix)
      for (int iy = 0; iy < 2; ++iy)
        for (int iz = 0; iz < 2; ++iz) k[c++] = m4_transform_point(s->query_from_clip, V3d{sg[ix], sg[iy], sg[iz]});
    for (int i = 0; i < 8; ++i) {
      s->corners[3 * i] = k[i].x;
      s->corners[3 * i + 1] = k[i].y;
      s->corners[3 * i + 2] = k[i].z;
    }
    V3d e[6], n[5];
    e[0] = v_normalize(v_sub(k[4], k[0]));
    e[1] = v_normalize(v_sub(k[2], k[0]));
    e[2] = v_normalize(v_sub(k[1], k[0]));
    e[3] = v_normalize(v_sub(k[3], k[2]));
    e[4] = v_normalize(v_sub(k[5], k[4]));
    e[5] = v_normalize(v_sub(k[7], k[6]));
    n[0] = v_normalize(v_cross(e[0], e[1]));
    n[1] = v_normalize(v_cross(e[0], e[2]));
    n[2] = v_normalize(v_cross(e[0], e[3]));
    n[3] = v_normalize(v_cross(e[1], e[2]));
    n[4] = v_normalize(v_cross(e[1], e[4]));
    cache_axes_for_aabb(s, e, 6, n, 5);
  } else if (s->kind == PCV_SHAPE_OBB) {
    // s->iso holds query_from_obb on entry; corners/edges use it, contains() needs the inverse (obb.rs:35-41)
    const double* q = s->iso + 3;
    V3d t = {s->iso[0], s->iso[1], s->iso[2]};
    const double sx[8] = {-1, 1, -1, 1, -1, 1, -1, 1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    for (int c = 0; c < 8; ++c) {
      V3d p = v_add(quat_rotate(q, V3d{sx[c] * s->half[0], sy[c] * s->half[1], sz[c] * s->half[2]}), t);
      s->corners[3 * c] = p.x;
      s->corners[3 * c + 1] = p.y;
      s->corners[3 * c + 2] = p.z;
    }
    V3d e[3];
    e[0] = v_normalize(quat_rotate(q, V3d{1, 0, 0}));
    e[1] = v_normalize(quat_rotate(q, V3d{0, 1, 0}));
    e[2] = v_normalize(quat_rotate(q, V3d{0, 0, 1}));
    cache_axes_for_aabb(s, e, 3, e, 3);
    double qi[4] = {-q[0], -q[1], -q[2], q[3]};  // Isometry3::inverse
    V3d ti = quat_rotate(qi, V3d{-t.x, -t.y, -t.z});
    s->iso[0] = ti.x;
    s->iso[1] = ti.y;
    s->iso[2] = ti.z;
    s->iso[3] = qi[0];
    s->iso[4] = qi[1];
    s->iso[5] = qi[2];
    s->iso[6] = qi[3];
  } else if (s->kind == PCV_SHAPE_AABB) {  // aabb.rs:98-125
    const double* mn = s->bmin;
    const double* mx = s->bmax;
    const double cs[24] = {mn[0], mn[1], mn[2], mx[0], mn[1], mn[2], mn[0], mx[1], mn[2], mx[0], mx[1], mn[2],
                           mn[0], mn[1], mx[2], mx[0], mn[1], mx[2], mn[0], mx[1], mx[2], mx[0], mx[1], mx[2]};
    for (int i = 0; i < 24; ++i) s->corners[i] = cs[i];
    const double ax[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) s->axes[i] = ax[i];
    s->naxes = 3;
  } else {
    s->naxes = 0;  // AllPoints
  }
  for (int a = 0; a < s->naxes; ++a)
    project8(s->corners, V3d{s->axes[3 * a], s->axes[3 * a + 1], s->axes[3 * a + 2]}, &s->amin[a], &s->amax[a]);
}

// sat() of one cube against one prepared shape. Evaluating every axis gives the same Relation as the reference's
// early return: Out if any axis separates, else Cross if B sticks out on any axis, else In (sat.rs:174-194).
__device__ __forceinline__ int sat_cube(const PcvShapeDev* __restrict__ s, double mnx, double mny, double mnz, double edge) {
  if (s->kind == PCV_SHAPE_ALL) return 1;  // AllPoints intersects everything (math/mod.rs:139-160) -> "not Out"
  // Cube::to_aabb: Aabb::new(min, min + edge) (inf / sup)
  const double ax_ = mnx + edge, ay_ = mny + edge, az_ = mnz + edge;
  const double lx = fmin(mnx, ax_), hx = fmax(mnx, ax_);
  const double ly = fmin(mny, ay_), hy = fmax(mny, ay_);
  const double lz = fmin(mnz, az_), hz = fmax(mnz, az_);
  bool out = false, cross = false;
  const int na = s->naxes;
  for (int a = 0; a < na; ++a) {
    const double ax = s->axes[3 * a], ay = s->axes[3 * a + 1], az = s->axes[3 * a + 2];
    const double plx = lx * ax, phx = hx * ax, ply = ly * ay, phy = hy * ay, plz = lz * az, phz = hz * az;
    // corners in aabb.rs:114-125 order: (l,l,l) (h,l,l) (l,h,l) (h,h,l) (l,l,h) (h,l,h) (l,h,h) (h,h,h)
    double c0 = (plx + ply) + plz, c1 = (phx + ply) + plz, c2 = (plx + phy) + plz, c3 = (phx + phy) + plz;
    double c4 = (plx + ply) + phz, c5 = (phx + ply) + phz, c6 = (plx + phy) + phz, c7 = (phx + phy) + phz;
    double bmin = fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(1.7976931348623157e308, c0), c1), c2), c3), c4), c5), c6), c7);
    double bmax = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(-1.7976931348623157e308, c0), c1), c2), c3), c4), c5), c6), c7);
    const double amin = s->amin[a], amax = s->amax[a];
    out = out || (bmin > amax || bmax < amin);
    cross = cross || (amin > bmin || bmax > amax);
  }
  return out ? 2 : (cross ? 1 : 0);
}

__device__ __forceinline__ double clamp_num(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// octree/mod.rs:103-139. NaN marks the cases where the reference panics (w == 0).
__device__ double size_on_screen(const double* __restrict__ m, double mnx, double mny, double mnz, double edge) {
  const double mxx = mnx + edge, mxy = mny + edge, mxz = mnz + edge;
  const double px[8] = {mnx, mxx, mxx, mnx, mxx, mnx, mxx, mnx};
  const double py[8] = {mny, mxy, mny, mxy, mxy, mny, mny, mxy};
  const double pz[8] = {mnz, mxz, mnz, mnz, mnz, mxz, mxz, mxz};
  double lox = 0, hix = 0, loy = 0, hiy = 0;
  bool bad = false;
  for (int i = 0; i < 8; ++i) {
    double v[4];
    for (int r = 0; r < 4; ++r) v[r] = ((M4(m, r, 0) * px[i] + M4(m, r, 1) * py[i]) + M4(m, r, 2) * pz[i]) + M4(m, r, 3) * 1.0;
    if (v[3] == 0.0) bad = true;
    const double cx = clamp_num(v[0] / v[3], -1., 1.), cy = clamp_num(v[1] / v[3], -1., 1.);
    if (i == 0) {
      lox = hix = cx;
      loy = hiy = cy;
    } else {
      lox = fmin(lox, cx);
      hix = fmax(hix, cx);
      loy = fmin(loy, cy);
      hiy = fmax(hiy, cy);
    }
  }
  if (bad) return __longlong_as_double(0x7ff8000000000000LL);
  return (hix - lox) * (hiy - loy);
}

// K7: grid.y = shape, grid.x covers the nodes.
__global__ __launch_bounds__(256) void cull_nodes_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t m,
                                                          const double* __restrict__ cubes /* m x 4 */,
                                                          uint8_t* __restrict__ relation, double* __restrict__ sizes) {
  const PcvShapeDev* s = shapes + blockIdx.y;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const double4 c = *reinterpret_cast<const double4*>(cubes + 4 * (uint64_t)i);
  const uint64_t o = (uint64_t)blockIdx.y * m + i;
  relation[o] = s->valid ? (uint8_t)sat_cube(s, c.x, c.y, c.z, c.w) : (uint8_t)2;
  if (sizes) sizes[o] = size_on_screen(s->clip_from_query, c.x, c.y, c.z, c.w);
}

struct QTree {
  uint32_t m;
  const double* cubes;         // get_child-style cubes (min xyz, edge), node order = (level, index)
  const uint32_t* first_child;
  const uint8_t* child_mask;
  const uint8_t* empty;        // num_points == 0
};

// K7b: one lane per frustum; the heap lives in global scratch (capacity m entries per frustum).
struct HeapEntry {
  double size;
  uint32_t node;
  uint32_t relation;
};
__device__ __forceinline__ void heap_sift_up(HeapEntry* d, uint32_t start, uint32_t pos) {
  HeapEntry elt = d[pos];
  while (pos > start) {
    uint32_t parent = (pos - 1) / 2;
    if (elt.size <= d[parent].size) break;
    d[pos] = d[parent];
    pos = parent;
  }
  d[pos] = elt;
}
__global__ __launch_bounds__(64) void visible_nodes_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t first_shape,
                                                            uint32_t nshapes, QTree t, HeapEntry* __restrict__ heaps,
                                                            uint32_t capacity, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ out, int32_t* __restrict__ status) {
  const uint32_t li = blockIdx.x * 64 + threadIdx.x;
  if (li >= nshapes) return;
  const uint32_t f = first_shape + li;
  const PcvShapeDev* s = shapes + f;
  HeapEntry* d = heaps + (uint64_t)li * t.m;
  uint32_t* o = out + (uint64_t)f * capacity;
  uint32_t len = 0, nout = 0;
  int32_t st = 0;
  if (!s->valid) {  // .expect("Invalid projection matrix.")
    counts[f] = 0;
    status[f] = 1;
    return;
  }
  if (t.m > 0) {  // maybe_push_node(root, Cross)
    double sz = size_on_screen(s->clip_from_query, t.cubes[0], t.cubes[1], t.cubes[2], t.cubes[3]);
    if (sz != sz) st = 2;
    d[0] = HeapEntry{sz, 0u, 1u};
    len = 1;
  }
  while (len > 0 && st == 0) {
    // BinaryHeap::pop
    HeapEntry item = d[len - 1];
    --len;
    if (len > 0) {
      HeapEntry top = d[0];
      d[0] = item;
      item = top;
      uint32_t end = len, pos = 0, child = 1;
      HeapEntry elt = d[0];
      while (child + 1 < end) {
        child += (d[child].size <= d[child + 1].size) ? 1u : 0u;
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
      }
      if (child == end - 1) {
        d[pos] = d[child];
        pos = child;
      }
      d[pos] = elt;
      heap_sift_up(d, 0, pos);
    }
    const uint32_t mask = t.child_mask[item.node];
    uint32_t cidx = t.first_child[item.node];
    for (uint32_t ci = 0; ci < 8; ++ci) {
      if (!((mask >> ci) & 1u)) continue;  // maybe_push_node: only nodes that exist
      const uint32_t c = cidx++;
      const double* cb = t.cubes + 4 * (uint64_t)c;
      uint32_t rel = 0;
      if (item.relation == 1u) {
        rel = (uint32_t)sat_cube(s, cb[0], cb[1], cb[2], cb[3]);
        if (rel == 2u) continue;
      }
      double sz = size_on_screen(s->clip_from_query, cb[0], cb[1], cb[2], cb[3]);
      if (sz != sz) st = 2;
      d[len] = HeapEntry{sz, c, rel};
      heap_sift_up(d, 0, len);
      ++len;
    }
    if (!t.empty[item.node]) {
      if (nout < capacity) o[nout] = item.node;
      ++nout;
    }
  }
  counts[f] = nout;
  status[f] = st;
}

// K7c: BFS of NodeIdsIterator; queue in global scratch.
__global__ __launch_bounds__(64) void nodes_in_location_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t first_shape,
                                                                uint32_t nshapes, QTree t, const double* __restrict__ fb_cubes,
                                                                uint32_t* __restrict__ queues, uint32_t capacity,
                                                                uint32_t* __restrict__ counts, uint32_t* __restrict__ out) {
  const uint32_t li = blockIdx.x * 64 + threadIdx.x;
  if (li >= nshapes) return;
  const uint32_t f = first_shape + li;
  const PcvShapeDev* s = shapes + f;
  uint32_t* q = queues + (uint64_t)li * t.m;
  uint32_t* o = out + (uint64_t)f * capacity;
  uint32_t head = 0, tail = 0, nout = 0;
  if (t.m > 0 && s->valid) q[tail++] = 0;
  while (head < tail) {
    const uint32_t cur = q[head++];
    const double* cb = fb_cubes + 4 * (uint64_t)cur;  // NodeMeta::bounding_cube = find_bounding_cube (octree/mod.rs:205)
    if (sat_cube(s, cb[0], cb[1], cb[2], cb[3]) == 2) continue;
    const uint32_t mask = t.child_mask[cur];
    uint32_t cidx = t.first_child[cur];
    for (uint32_t ci = 0; ci < 8; ++ci)
      if ((mask >> ci) & 1u) q[tail++] = cidx++;
    if (nout < capacity) o[nout] = cur;
    ++nout;
  }
  counts[f] = nout;
}

// K8: keep mask. Positions either raw f64 SoA or a node's encoded bytes.
struct PointsView {
  uint64_t n;
  const double *x, *y, *z;
  const uint8_t* encoded;  // non-null: node bytes, `enc`, cube
  uint32_t enc;
  double cube_min[3];
  double cube_edge;
  const float* attr;       // optional f32 attribute with closed interval
  double lo, hi;
  int has_interval;
};

__device__ __forceinline__ V3d load_point(const PointsView& v, uint64_t i) {
  if (!v.encoded) return {v.x[i], v.y[i], v.z[i]};
  uint64_t c[3];
  switch (v.enc) {
    case PCV_ENC_UINT8: {
      const uint8_t* p = v.encoded + 3 * i;
      c[0] = p[0];
      c[1] = p[1];
      c[2] = p[2];
      break;
    }
    case PCV_ENC_UINT16: {
      const uint16_t* p = reinterpret_cast<const uint16_t*>(v.encoded) + 3 * i;
      c[0] = p[0];
      c[1] = p[1];
      c[2] = p[2];
      break;
    }
    case PCV_ENC_FLOAT32: {
      const uint32_t* p = reinterpret_cast<const uint32_t*>(v.encoded) + 3 * i;
      c[0] = p[0];
      c[1] = p[1];
      c[2] = p[2];
      break;
    }
    default: {
      const uint64_t* p = reinterpret_cast<const uint64_t*>(v.encoded) + 3 * i;
      c[0] = p[0];
      c[1] = p[1];
      c[2] = p[2];
      break;
    }
  }
  return {pcv_decode_coord(v.enc, c[0], v.cube_min[0], v.cube_edge), pcv_decode_coord(v.enc, c[1], v.cube_min[1], v.cube_edge),
          pcv_decode_coord(v.enc, c[2], v.cube_min[2], v.cube_edge)};
}

__global__ __launch_bounds__(256) void cull_points_kernel(const PcvShapeDev* __restrict__ shape, PointsView v,
                                                           uint8_t* __restrict__ keep, unsigned long long* __restrict__ kept) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool k = false;
  if (i < v.n) {
    const V3d p = load_point(v, i);
    switch (shape->kind) {
      case PCV_SHAPE_AABB:  // aabb.rs:46-48: mins <= p < maxs
        k = shape->bmin[0] <= p.x && shape->bmin[1] <= p.y && shape->bmin[2] <= p.z && p.x < shape->bmax[0] &&
            p.y < shape->bmax[1] && p.z < shape->bmax[2];
        break;
      case PCV_SHAPE_FRUSTUM:
      case PCV_SHAPE_FRUSTUM_WITH_INVERSE: {  // frustum.rs:120-125
        const V3d c = m4_transform_point(shape->clip_from_query, p);
        const double mn = fmin(fmin(c.x, c.y), c.z), mx = fmax(fmax(c.x, c.y), c.z);
        k = mn > -1.0 && mx < 1.0;
        break;
      }
      case PCV_SHAPE_OBB: {  // obb.rs:83-90
        const V3d q = v_add(quat_rotate(shape->iso + 3, p), V3d{shape->iso[0], shape->iso[1], shape->iso[2]});
        k = fabs(q.x) <= shape->half[0] && fabs(q.y) <= shape->half[1] && fabs(q.z) <= shape->half[2];
        break;
      }
      default: k = true;  // AllPoints
    }
    if (v.has_interval) {  // iterator.rs:82-91 + math/mod.rs:86-88
      const double a = (double)v.attr[i];
      k = k && (v.lo <= a && a <= v.hi);
    }
    keep[i] = k ? 1 : 0;
  }
  const unsigned long long b = __ballot(k);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(kept, (unsigned long long)__popcll(b));
}

// K9
__global__ __launch_bounds__(256) void transform_points_kernel(uint64_t n, const double* __restrict__ x,
                                                                const double* __restrict__ y, const double* __restrict__ z,
                                                                double t0, double t1, double t2, double q0, double q1,
                                                                double q2, double q3, double* __restrict__ ox,
                                                                double* __restrict__ oy, double* __restrict__ oz) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double q[4] = {q0, q1, q2, q3};
  const V3d r = v_add(quat_rotate(q, V3d{x[i], y[i], z[i]}), V3d{t0, t1, t2});
  ox[i] = r.x;
  oy[i] = r.y;
  oz[i] = r.z;
}

// ---- batched point query (the work of ParallelIterator + FilteredIterator for one location) -------
struct QueryJob {
  uint64_t xyz_off;    // byte offset of the node's encoded positions in the xyz blob
  uint64_t point_off;  // point offset in the rgb / intensity blobs
  uint64_t first;      // index of the node's first point in the concatenated job space
  uint32_t n;
  uint32_t enc;
  double cube_min[3];
  double cube_edge;
};

__device__ __forceinline__ uint32_t find_job(const QueryJob* __restrict__ jobs, uint32_t njobs, uint64_t i) {
  uint32_t lo = 0, hi = njobs;  // last job with first <= i
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobs[mid].first <= i) lo = mid;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ V3d job_point(const QueryJob& jb, const uint8_t* __restrict__ xyz_blob, uint64_t k) {
  PointsView v{};
  v.encoded = xyz_blob + jb.xyz_off;
  v.enc = jb.enc;
  v.cube_min[0] = jb.cube_min[0];
  v.cube_min[1] = jb.cube_min[1];
  v.cube_min[2] = jb.cube_min[2];
  v.cube_edge = jb.cube_edge;
  return load_point(v, k);
}

__device__ __forceinline__ bool shape_contains(const PcvShapeDev* __restrict__ shape, V3d p) {
  switch (shape->kind) {
    case PCV_SHAPE_AABB:
      return shape->bmin[0] <= p.x && shape->bmin[1] <= p.y && shape->bmin[2] <= p.z && p.x < shape->bmax[0] &&
             p.y < shape->bmax[1] && p.z < shape->bmax[2];
    case PCV_SHAPE_FRUSTUM:
    case PCV_SHAPE_FRUSTUM_WITH_INVERSE: {
      const V3d c = m4_transform_point(shape->clip_from_query, p);
      const double mn = fmin(fmin(c.x, c.y), c.z), mx = fmax(fmax(c.x, c.y), c.z);
      return mn > -1.0 && mx < 1.0;
    }
    case PCV_SHAPE_OBB: {
      const V3d q = v_add(quat_rotate(shape->iso + 3, p), V3d{shape->iso[0], shape->iso[1], shape->iso[2]});
      return fabs(q.x) <= shape->half[0] && fabs(q.y) <= shape->half[1] && fabs(q.z) <= shape->half[2];
    }
    default: return true;
  }
}

// pass 1: keep flag per point of every job + kept count per workgroup
__global__ __launch_bounds__(256) void query_flags_kernel(const PcvShapeDev* __restrict__ shape,
                                                           const QueryJob* __restrict__ jobs, uint32_t njobs, uint64_t total,
                                                           const uint8_t* __restrict__ xyz_blob,
                                                           const float* __restrict__ inten_blob, int has_interval, double lo,
                                                           double hi, uint8_t* __restrict__ keep,
                                                           uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wave_cnt[4];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool k = false;
  if (i < total) {
    const QueryJob jb = jobs[find_job(jobs, njobs, i)];
    const uint64_t kk = i - jb.first;
    k = shape_contains(shape, job_point(jb, xyz_blob, kk));
    if (has_interval) {
      const double a = (double)inten_blob[jb.point_off + kk];
      k = k && (lo <= a && a <= hi);
    }
    keep[i] = k ? 1 : 0;
  }
  const unsigned long long b = __ballot(k);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// exclusive scan of nb counters by one workgroup; total to out_total
__global__ __launch_bounds__(1024) void query_scan_kernel(uint32_t* __restrict__ counts, uint32_t nb,
                                                           unsigned long long* __restrict__ out_total) {
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t idx = base + threadIdx.x;
    const unsigned long long v = idx < nb ? counts[idx] : 0ull;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      woff += (w < wave) ? wave_tot[w] : 0ull;
      tot += wave_tot[w];
    }
    // positions fit u32 per octree (n < 2^32); keep the exclusive prefix in place
    if (idx < nb) counts[idx] = (uint32_t)(running + woff + inc - v);
    __syncthreads();
    if (threadIdx.x == 0) running += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_total = running;
}

// pass 2: stable compaction — decoded f64 positions, colours and intensity of the kept points, in job order
__global__ __launch_bounds__(256) void query_compact_kernel(const QueryJob* __restrict__ jobs, uint32_t njobs, uint64_t total,
                                                             const uint8_t* __restrict__ xyz_blob,
                                                             const uint8_t* __restrict__ rgb_blob,
                                                             const float* __restrict__ inten_blob,
                                                             const uint8_t* __restrict__ keep,
                                                             const uint32_t* __restrict__ block_offsets, uint64_t capacity,
                                                             double* __restrict__ ox, double* __restrict__ oy,
                                                             double* __restrict__ oz, uint8_t* __restrict__ orgb,
                                                             float* __restrict__ ointen) {
  __shared__ uint32_t wave_cnt[4];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool k = i < total && keep[i];
  const unsigned long long b = __ballot(k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  if (!k) return;
  uint32_t pos = block_offsets[blockIdx.x] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
  if (pos >= capacity) return;
  const QueryJob jb = jobs[find_job(jobs, njobs, i)];
  const uint64_t kk = i - jb.first;
  const V3d p = job_point(jb, xyz_blob, kk);
  ox[pos] = p.x;
  oy[pos] = p.y;
  oz[pos] = p.z;
  const uint8_t* c = rgb_blob + 3 * (jb.point_off + kk);
  orgb[3 * (uint64_t)pos] = c[0];
  orgb[3 * (uint64_t)pos + 1] = c[1];
  orgb[3 * (uint64_t)pos + 2] = c[2];
  if (ointen) ointen[pos] = inten_blob[jb.point_off + kk];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
extern "C" int pcv_shapes_create(pcv_ctx* ctx, const pcv_shape* shapes, uint32_t count, pcv_shapes** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out || (count && !shapes)) return ctx->fail(PCV_E_INVALID, "null argument");
  *out = nullptr;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<PcvShapeDev> h(count);
  for (uint32_t i = 0; i < count; ++i) {
    const pcv_shape& s = shapes[i];
    PcvShapeDev& d = h[i];
    std::memset(&d, 0, sizeof(d));
    d.kind = s.kind;
    switch (s.kind) {
      case PCV_SHAPE_ALL: break;
      case PCV_SHAPE_AABB:
        for (int a = 0; a < 3; ++a) {  // Aabb::new: inf / sup of the two corners (aabb.rs:21-26)
          d.bmin[a] = std::fmin(s.params[a], s.params[3 + a]);
          d.bmax[a] = std::fmax(s.params[a], s.params[3 + a]);
        }
        break;
      case PCV_SHAPE_FRUSTUM:
        for (int a = 0; a < 16; ++a) d.clip_from_query[a] = s.params[a];
        break;
      case PCV_SHAPE_FRUSTUM_WITH_INVERSE:
        for (int a = 0; a < 16; ++a) {
          d.clip_from_query[a] = s.params[a];
          d.query_from_clip[a] = s.params[16 + a];
        }
        break;
      case PCV_SHAPE_OBB:
        for (int a = 0; a < 7; ++a) d.iso[a] = s.params[a];
        for (int a = 0; a < 3; ++a) d.half[a] = s.params[7 + a];
        break;
      default: return ctx->fail(PCV_E_INVALID, "unknown shape kind");
    }
  }
  pcv_shapes* r = new pcv_shapes();
  r->ctx = ctx;
  r->count = count;
  r->dev = nullptr;
  void* p = nullptr;
  int rc = ctx->dev_alloc(&p, sizeof(PcvShapeDev) * (count ? count : 1));
  if (rc) {
    delete r;
    return rc;
  }
  r->dev = (PcvShapeDev*)p;
  if (count) {
    hipError_t e = hipMemcpyAsync(r->dev, h.data(), sizeof(PcvShapeDev) * count, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(shape_setup_kernel, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, r->dev, count);
      e = hipStreamSynchronize(ctx->stream);  // `h` must outlive the copy
    }
    if (e != hipSuccess) {
      ctx->dev_free(r->dev);
      delete r;
      return ctx->fail(PCV_E_HIP, hipGetErrorString(e));
    }
  }
  *out = r;
  return PCV_OK;
}

extern "C" void pcv_shapes_free(pcv_shapes* s) {
  if (!s) return;
  s->ctx->dev_free(s->dev);
  delete s;
}

extern "C" uint32_t pcv_shapes_count(const pcv_shapes* s) { return s ? s->count : 0; }

extern "C" int pcv_shapes_get(pcv_shapes* s, uint32_t i, double corners[24], double axes[78], uint32_t* num_axes,
                              int* valid) {
  if (!s || i >= s->count) return PCV_E_INVALID;
  pcv_ctx* ctx = s->ctx;
  PcvShapeDev h;
  PCV_HIP_CHECK(ctx, hipMemcpy(&h, s->dev + i, sizeof(h), hipMemcpyDeviceToHost));
  if (corners) std::memcpy(corners, h.corners, sizeof(h.corners));
  if (axes) std::memcpy(axes, h.axes, sizeof(h.axes));
  if (num_axes) *num_axes = (uint32_t)h.naxes;
  if (valid) *valid = h.valid;
  return PCV_OK;
}

// Device-resident query view of an octree, built lazily (pcv_octree::query).
struct PcvOctreeQuery {
  uint32_t m = 0;
  double* cubes = nullptr;     // Node::get_child recurrence
  double* fb_cubes = nullptr;  // NodeId::find_bounding_cube recurrence
  uint32_t* first_child = nullptr;
  uint8_t* child_mask = nullptr;
  uint8_t* empty = nullptr;
  std::vector<uint32_t> h_first_child;  // host copies for host-side traversals
  std::vector<uint8_t> h_child_mask;
};

int pcv_octree_prepare_query(pcv_octree* t) {
  if (t->query) return PCV_OK;
  pcv_ctx* ctx = t->ctx;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const uint32_t m = (uint32_t)t->nodes.size();
  std::vector<double> cubes(4 * (size_t)m + 4), fb(4 * (size_t)m + 4);
  std::vector<uint32_t> first(m + 1, 0);
  std::vector<uint8_t> mask(m + 1, 0), empty(m + 1, 0);
  // nodes are sorted by (level, index): children of a node are contiguous, in digit order
  typedef unsigned __int128 u128;
  auto idx_of = [&](const pcv_node_info& n) { return ((u128)(n.id_high & 0x00ffffffffffffffull) << 64) | n.id_low; };
  std::vector<uint32_t> level_start(258, m);
  for (uint32_t i = m; i-- > 0;) level_start[t->nodes[i].level] = i;
  for (int l = 255; l >= 0; --l)
    if (level_start[l] == m && l + 1 < 258) level_start[l] = level_start[l + 1];
  // root cube: Cube::bounding (aabb.rs:149-157)
  const double root_edge =
      std::fmax(std::fmax(t->bbox_max[0] - t->bbox_min[0], t->bbox_max[1] - t->bbox_min[1]), t->bbox_max[2] - t->bbox_min[2]);
  std::vector<int> has_parent(m, 0);
  for (uint32_t i = 0; i < m; ++i) {
    const pcv_node_info& n = t->nodes[i];
    empty[i] = n.num_points == 0;
    for (int a = 0; a < 3; ++a) fb[4 * (size_t)i + a] = n.cube_min[a];
    fb[4 * (size_t)i + 3] = n.cube_edge;
    if (n.level == 0) {
      for (int a = 0; a < 3; ++a) cubes[a] = t->bbox_min[a];
      cubes[3] = root_edge;
      has_parent[i] = 1;
    }
    // children: binary search the next level for index * 8 .. index * 8 + 7
    const uint32_t lo = level_start[n.level + 1], hi = level_start[n.level + 2];
    const u128 want = idx_of(n) << 3;
    uint32_t a = lo, b = hi;
    while (a < b) {
      uint32_t mid = a + (b - a) / 2;
      if (idx_of(t->nodes[mid]) < want) a = mid + 1;
      else b = mid;
    }
    first[i] = a;
    uint32_t c = a;
    while (c < hi && (idx_of(t->nodes[c]) >> 3) == idx_of(n) && t->nodes[c].level == n.level + 1) {
      const unsigned digit = (unsigned)(idx_of(t->nodes[c]) & 7);
      mask[i] |= (uint8_t)(1u << digit);
      if (has_parent[i]) {  // Node::get_child (node.rs:190-211): min += half only where the bit is set
        const double half = cubes[4 * (size_t)i + 3] / 2.;
        double* cc = &cubes[4 * (size_t)c];
        cc[0] = cubes[4 * (size_t)i + 0];
        cc[1] = cubes[4 * (size_t)i + 1];
        cc[2] = cubes[4 * (size_t)i + 2];
        if (digit & 1) cc[2] += half;
        if (digit & 2) cc[1] += half;
        if (digit & 4) cc[0] += half;
        cc[3] = half;
        has_parent[c] = 1;
      }
      ++c;
    }
  }
  PcvOctreeQuery* q = new PcvOctreeQuery();
  q->m = m;
  q->h_first_child = first;
  q->h_child_mask = mask;
  void* p;
  int rc;
  size_t bytes = (size_t)(m + 1) * (64 + 4 + 2);
  if ((rc = ctx->dev_alloc(&p, bytes))) {
    delete q;
    return rc;
  }
  uint8_t* base = (uint8_t*)p;
  q->cubes = (double*)base;
  q->fb_cubes = q->cubes + 4 * (size_t)(m + 1);
  q->first_child = (uint32_t*)(q->fb_cubes + 4 * (size_t)(m + 1));
  q->child_mask = (uint8_t*)(q->first_child + (m + 1));
  q->empty = q->child_mask + (m + 1);
  hipError_t e = hipMemcpy(q->cubes, cubes.data(), 32 * (size_t)m, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(q->fb_cubes, fb.data(), 32 * (size_t)m, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(q->first_child, first.data(), 4 * (size_t)m, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(q->child_mask, mask.data(), (size_t)m, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(q->empty, empty.data(), (size_t)m, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    ctx->dev_free(p);
    delete q;
    return ctx->fail(PCV_E_HIP, hipGetErrorString(e));
  }
  t->query = q;
  return PCV_OK;
}

void pcv_octree_release_query(pcv_octree* t) {
  if (!t->query) return;
  t->ctx->dev_free(t->query->cubes);
  delete t->query;
  t->query = nullptr;
}

extern "C" int pcv_cull_nodes(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint8_t* relation,
                              double* size_on_screen_out) {
  if (!ctx) return PCV_E_INVALID;
  if (!shapes || !tree || !relation) return ctx->fail(PCV_E_INVALID, "null argument");
  int rc = pcv_octree_prepare_query(tree);
  if (rc) return rc;
  const uint32_t m = tree->query->m, f = shapes->count;
  if (m == 0 || f == 0) return PCV_OK;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  uint8_t* d_rel;
  double* d_sz = nullptr;
  if ((rc = sc.get(&d_rel, (size_t)f * m))) return rc;
  if (size_on_screen_out && (rc = sc.get(&d_sz, (size_t)f * m))) return rc;
  {
    PcvProf prof(ctx, PCV_K_CULL_NODES);
    // cull against NodeMeta cubes (find_bounding_cube), as nodes_in_location does; get_visible_nodes' own
    // get_child cubes differ at most in the sign of zero
    hipLaunchKernelGGL(cull_nodes_kernel, dim3((m + 255) / 256, f), dim3(256), 0, ctx->stream, shapes->dev, m,
                       tree->query->fb_cubes, d_rel, d_sz);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(relation, d_rel, (size_t)f * m, hipMemcpyDeviceToHost, ctx->stream));
  if (d_sz) PCV_HIP_CHECK(ctx, hipMemcpyAsync(size_on_screen_out, d_sz, (size_t)f * m * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_resolve();
  return PCV_OK;
}

static int traverse(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint32_t capacity, uint32_t* counts,
                    uint32_t* node_indices, int32_t* status, bool visible) {
  if (!ctx) return PCV_E_INVALID;
  if (!shapes || !tree || !counts || (capacity && !node_indices)) return ctx->fail(PCV_E_INVALID, "null argument");
  int rc = pcv_octree_prepare_query(tree);
  if (rc) return rc;
  const uint32_t m = tree->query->m, f = shapes->count;
  if (f == 0) return PCV_OK;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  uint32_t *d_counts, *d_out;
  int32_t* d_status;
  if ((rc = sc.get(&d_counts, f)) || (rc = sc.get(&d_out, (size_t)f * (capacity ? capacity : 1))) || (rc = sc.get(&d_status, f))) return rc;
  PCV_HIP_CHECK(ctx, hipMemsetAsync(d_status, 0, 4 * (size_t)f, ctx->stream));
  // scratch per in-flight shape: m heap entries (16 B) or m queue slots (4 B); bound it to ~256 MiB per batch
  const size_t per = visible ? sizeof(HeapEntry) * (size_t)(m ? m : 1) : 4 * (size_t)(m ? m : 1);
  uint32_t batch = (uint32_t)std::min<size_t>(f, std::max<size_t>(64, ((size_t)256 << 20) / per));
  void* scratch;
  if ((rc = ctx->dev_alloc(&scratch, per * batch))) return rc;
  sc.ptrs.push_back(scratch);
  QTree qt{m, tree->query->cubes, tree->query->first_child, tree->query->child_mask, tree->query->empty};
  for (uint32_t first = 0; first < f; first += batch) {
    const uint32_t nb = std::min(batch, f - first);
    PcvProf prof(ctx, visible ? PCV_K_VISIBLE_NODES : PCV_K_NODES_IN_LOCATION);
    if (visible)
      hipLaunchKernelGGL(visible_nodes_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, shapes->dev, first, nb, qt,
                         (HeapEntry*)scratch, capacity, d_counts, d_out, d_status);
    else
      hipLaunchKernelGGL(nodes_in_location_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, shapes->dev, first, nb,
                         qt, tree->query->fb_cubes, (uint32_t*)scratch, capacity, d_counts, d_out);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(counts, d_counts, 4 * (size_t)f, hipMemcpyDeviceToHost, ctx->stream));
  if (capacity) PCV_HIP_CHECK(ctx, hipMemcpyAsync(node_indices, d_out, 4 * (size_t)f * capacity, hipMemcpyDeviceToHost, ctx->stream));
  if (status) PCV_HIP_CHECK(ctx, hipMemcpyAsync(status, d_status, 4 * (size_t)f, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_resolve();
  return PCV_OK;
}

extern "C" int pcv_visible_nodes(pcv_ctx* ctx, const pcv_shapes* frusta, pcv_octree* tree, uint32_t capacity,
                                 uint32_t* counts, uint32_t* node_indices, int32_t* status) {
  return traverse(ctx, frusta, tree, capacity, counts, node_indices, status, true);
}
extern "C" int pcv_nodes_in_location(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint32_t capacity,
                                     uint32_t* counts, uint32_t* node_indices) {
  return traverse(ctx, shapes, tree, capacity, counts, node_indices, nullptr, false);
}

static int run_cull_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, PointsView v, const float* attr,
                           const double* interval, int mem, uint8_t* keep, uint64_t* kept) {
  if (!shapes || shape_index >= shapes->count || !keep) return ctx->fail(PCV_E_INVALID, "bad shape / null output");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  int rc;
  if (kept) *kept = 0;
  if (v.n == 0) return PCV_OK;
  if (interval) {
    if (!attr) return ctx->fail(PCV_E_INVALID, "interval without attribute");
    v.has_interval = 1;
    v.lo = interval[0];
    v.hi = interval[1];
    if (mem == PCV_MEM_HOST) {
      float* da;
      if ((rc = sc.get(&da, v.n))) return rc;
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(da, attr, v.n * 4, hipMemcpyHostToDevice, ctx->stream));
      v.attr = da;
    } else {
      v.attr = attr;
    }
  }
  uint8_t* d_keep = keep;
  if (mem == PCV_MEM_HOST && (rc = sc.get(&d_keep, v.n))) return rc;
  unsigned long long* d_cnt;
  if ((rc = sc.get(&d_cnt, 1))) return rc;
  PCV_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_CULL_POINTS);
    hipLaunchKernelGGL(cull_points_kernel, dim3((unsigned)((v.n + 255) / 256)), dim3(256), 0, ctx->stream,
                       shapes->dev + shape_index, v, d_keep, d_cnt);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  if (mem == PCV_MEM_HOST) PCV_HIP_CHECK(ctx, hipMemcpyAsync(keep, d_keep, v.n, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_resolve();
  if (kept) *kept = ctx->mailbox[0];
  return PCV_OK;
}

extern "C" int pcv_cull_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, const pcv_points* points,
                               const double* interval, uint8_t* keep, uint64_t* kept) {
  if (!ctx) return PCV_E_INVALID;
  if (!points) return ctx->fail(PCV_E_INVALID, "points is null");
  if (points->n > 0 && (!points->x || !points->y || !points->z)) return ctx->fail(PCV_E_INVALID, "x/y/z must be non-null");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  PointsView v{};
  v.n = points->n;
  v.x = points->x;
  v.y = points->y;
  v.z = points->z;
  int rc;
  if (points->mem == PCV_MEM_HOST && points->n) {
    double *x, *y, *z;
    if ((rc = sc.get(&x, v.n)) || (rc = sc.get(&y, v.n)) || (rc = sc.get(&z, v.n))) return rc;
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(x, points->x, v.n * 8, hipMemcpyHostToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(y, points->y, v.n * 8, hipMemcpyHostToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(z, points->z, v.n * 8, hipMemcpyHostToDevice, ctx->stream));
    v.x = x;
    v.y = y;
    v.z = z;
  }
  return run_cull_points(ctx, shapes, shape_index, v, points->intensity, interval, points->mem, keep, kept);
}

extern "C" int pcv_cull_node_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree,
                                    uint64_t node, const double* interval, uint8_t* keep, uint64_t* kept) {
  if (!ctx) return PCV_E_INVALID;
  if (!tree || node >= tree->nodes.size()) return ctx->fail(PCV_E_INVALID, "bad node");
  if (!tree->d_xyz) {  // an octree opened from a directory: node files are uploaded on first use
    int lrc = pcv_octree_load_device(tree);
    if (lrc) return lrc;
  }
  const pcv_node_info& n = tree->nodes[node];
  PointsView v{};
  v.n = (uint64_t)n.num_points;
  v.encoded = tree->d_xyz + n.xyz_offset;
  v.enc = n.encoding;
  for (int a = 0; a < 3; ++a) v.cube_min[a] = n.cube_min[a];
  v.cube_edge = n.cube_edge;
  const float* attr = tree->has_intensity ? reinterpret_cast<const float*>(tree->d_int) + n.point_offset : nullptr;
  if (interval && !attr) return ctx->fail(PCV_E_INVALID, "octree has no intensity attribute to filter on");
  // keep is a HOST buffer here; the attribute already lives on the device
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  if (kept) *kept = 0;
  if (v.n == 0) return PCV_OK;
  if (!shapes || shape_index >= shapes->count || !keep) return ctx->fail(PCV_E_INVALID, "bad shape / null output");
  if (interval) {
    v.has_interval = 1;
    v.lo = interval[0];
    v.hi = interval[1];
    v.attr = attr;
  }
  uint8_t* d_keep;
  unsigned long long* d_cnt;
  int rc;
  if ((rc = sc.get(&d_keep, v.n)) || (rc = sc.get(&d_cnt, 1))) return rc;
  PCV_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_CULL_POINTS);
    hipLaunchKernelGGL(cull_points_kernel, dim3((unsigned)((v.n + 255) / 256)), dim3(256), 0, ctx->stream,
                       shapes->dev + shape_index, v, d_keep, d_cnt);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(keep, d_keep, v.n, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_resolve();
  if (kept) *kept = ctx->mailbox[0];
  return PCV_OK;
}

extern "C" int pcv_transform_points(pcv_ctx* ctx, const double iso[7], const pcv_points* points, double* ox, double* oy,
                                    double* oz) {
  if (!ctx) return PCV_E_INVALID;
  if (!iso || !points || !ox || !oy || !oz) return ctx->fail(PCV_E_INVALID, "null argument");
  const uint64_t n = points->n;
  if (n == 0) return PCV_OK;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  const double *x = points->x, *y = points->y, *z = points->z;
  double *dx = ox, *dy = oy, *dz = oz;
  int rc;
  if (points->mem == PCV_MEM_HOST) {
    double *ix, *iy, *iz;
    if ((rc = sc.get(&ix, n)) || (rc = sc.get(&iy, n)) || (rc = sc.get(&iz, n)) || (rc = sc.get(&dx, n)) ||
        (rc = sc.get(&dy, n)) || (rc = sc.get(&dz, n)))
      return rc;
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(ix, x, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(iy, y, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(iz, z, n * 8, hipMemcpyHostToDevice, ctx->stream));
    x = ix;
    y = iy;
    z = iz;
  }
  {
    PcvProf prof(ctx, PCV_K_TRANSFORM_POINTS);
    hipLaunchKernelGGL(transform_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, x, y, z,
                       iso[0], iso[1], iso[2], iso[3], iso[4], iso[5], iso[6], dx, dy, dz);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  if (points->mem == PCV_MEM_HOST) {
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(ox, dx, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(oy, dy, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(oz, dz, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_resolve();
  return PCV_OK;
}

// N3: nodes_in_location + per-point culling + stable compaction for one location in a handful of launches.
// only_node == nullptr: every node PointCloud::nodes_in_location reports for the shape; otherwise that one node
// (stream_points_for_query_in_node, src/iterator.rs:185-205: the node's points through the FilteredIterator)
static int query_points_impl(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree,
                             const uint64_t* only_node, const double* interval, uint64_t capacity, int mem, double* x, double* y,
                             double* z, uint8_t* rgb, float* intensity, uint64_t* count) {
  if (!ctx) return PCV_E_INVALID;
  if (!shapes || shape_index >= shapes->count || !tree || !count) return ctx->fail(PCV_E_INVALID, "bad argument");
  if (capacity && (!x || !y || !z || !rgb)) return ctx->fail(PCV_E_INVALID, "null output");
  if (mem != PCV_MEM_HOST && mem != PCV_MEM_DEVICE) return ctx->fail(PCV_E_INVALID, "bad mem");
  *count = 0;
  if (tree->nodes.empty()) return PCV_OK;
  if (!tree->d_xyz) {  // an octree opened from a directory: node files are uploaded on first use
    int lrc = pcv_octree_load_device(tree);
    if (lrc) return lrc;
  }
  if (interval && !tree->has_intensity) return ctx->fail(PCV_E_INVALID, "octree has no intensity attribute to filter on");
  int rc = pcv_octree_prepare_query(tree);
  if (rc) return rc;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  const uint32_t m = tree->query->m;
  std::vector<uint32_t> nodes;
  if (only_node) {
    if (*only_node >= tree->nodes.size()) return ctx->fail(PCV_E_INVALID, "bad node");
    nodes.push_back((uint32_t)*only_node);
  } else {
  // 1. PointCloud::nodes_in_location for this one shape: the Relation of every node cube in one dense launch
  //    (same sat() as the traversal kernel), then the breadth-first walk of NodeIdsIterator on the host.
  uint8_t* d_rel;
  if ((rc = sc.get(&d_rel, m))) return rc;
  {
    PcvProf prof(ctx, PCV_K_CULL_NODES);
    hipLaunchKernelGGL(cull_nodes_kernel, dim3((m + 255) / 256, 1), dim3(256), 0, ctx->stream, shapes->dev + shape_index, m,
                       tree->query->fb_cubes, d_rel, (double*)nullptr);
  }
  std::vector<uint8_t> rel(m);
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(rel.data(), d_rel, m, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  nodes.reserve(m);
  {
    std::vector<uint32_t> queue;
    queue.reserve(m);
    queue.push_back(0);
    for (size_t head = 0; head < queue.size(); ++head) {
      const uint32_t cur = queue[head];
      if (rel[cur] == 2) continue;
      uint32_t c = tree->query->h_first_child[cur];
      for (int ci = 0; ci < 8; ++ci)
        if ((tree->query->h_child_mask[cur] >> ci) & 1) queue.push_back(c++);
      nodes.push_back(cur);
    }
  }
  }
  const uint32_t nn = (uint32_t)nodes.size();
  // 2. one job per non-empty node, in traversal order
  std::vector<QueryJob> jobs;
  uint64_t total = 0;
  for (uint32_t k = 0; k < nn; ++k) {
    const pcv_node_info& nd = tree->nodes[nodes[k]];
    if (nd.num_points <= 0) continue;
    QueryJob jb;
    jb.xyz_off = nd.xyz_offset;
    jb.point_off = nd.point_offset;
    jb.first = total;
    jb.n = (uint32_t)nd.num_points;
    jb.enc = nd.encoding;
    for (int a = 0; a < 3; ++a) jb.cube_min[a] = nd.cube_min[a];
    jb.cube_edge = nd.cube_edge;
    jobs.push_back(jb);
    total += (uint64_t)nd.num_points;
  }
  if (total == 0) return PCV_OK;
  const uint32_t njobs = (uint32_t)jobs.size();
  const uint32_t nb = (uint32_t)((total + 255) / 256);
  QueryJob* d_jobs;
  uint8_t* d_keep;
  uint32_t* d_bc;
  unsigned long long* d_total;
  if ((rc = sc.get(&d_jobs, njobs)) || (rc = sc.get(&d_keep, total)) || (rc = sc.get(&d_bc, nb)) || (rc = sc.get(&d_total, 1)))
    return rc;
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(QueryJob) * njobs, hipMemcpyHostToDevice, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_CULL_POINTS);
    hipLaunchKernelGGL(query_flags_kernel, dim3(nb), dim3(256), 0, ctx->stream, shapes->dev + shape_index, d_jobs, njobs, total,
                       tree->d_xyz, (const float*)tree->d_int, interval ? 1 : 0, interval ? interval[0] : 0.0,
                       interval ? interval[1] : 0.0, d_keep, d_bc);
  }
  hipLaunchKernelGGL(query_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_bc, nb, d_total);
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // also keeps `jobs` alive until the copy is done
  const unsigned long long kept = ctx->mailbox[0];
  *count = kept;
  const uint64_t nout = kept < capacity ? kept : capacity;
  if (nout) {
    double *dx = x, *dy = y, *dz = z;
    uint8_t* drgb = rgb;
    float* dint = intensity;
    const bool want_int = intensity != nullptr && tree->has_intensity;
    if (mem == PCV_MEM_HOST) {
      if ((rc = sc.get(&dx, nout)) || (rc = sc.get(&dy, nout)) || (rc = sc.get(&dz, nout)) || (rc = sc.get(&drgb, 3 * nout))) return rc;
      if (want_int && (rc = sc.get(&dint, nout))) return rc;
    }
    {
      PcvProf prof(ctx, PCV_K_QUERY_COMPACT);
      hipLaunchKernelGGL(query_compact_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_jobs, njobs, total, tree->d_xyz,
                         tree->d_rgb, (const float*)tree->d_int, d_keep, d_bc, nout, dx, dy, dz, drgb, want_int ? dint : nullptr);
    }
    PCV_HIP_CHECK(ctx, hipGetLastError());
    if (mem == PCV_MEM_HOST) {
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(x, dx, 8 * nout, hipMemcpyDeviceToHost, ctx->stream));
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(y, dy, 8 * nout, hipMemcpyDeviceToHost, ctx->stream));
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(z, dz, 8 * nout, hipMemcpyDeviceToHost, ctx->stream));
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(rgb, drgb, 3 * nout, hipMemcpyDeviceToHost, ctx->stream));
      if (want_int) PCV_HIP_CHECK(ctx, hipMemcpyAsync(intensity, dint, 4 * nout, hipMemcpyDeviceToHost, ctx->stream));
    }
    PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->prof_resolve();
  return PCV_OK;
}

extern "C" int pcv_query_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree,
                                const double* interval, uint64_t capacity, int mem, double* x, double* y, double* z,
                                uint8_t* rgb, float* intensity, uint64_t* count) {
  return query_points_impl(ctx, shapes, shape_index, tree, nullptr, interval, capacity, mem, x, y, z, rgb, intensity, count);
}
extern "C" int pcv_query_node_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree,
                                     uint64_t node, const double* interval, uint64_t capacity, int mem, double* x, double* y,
                                     double* z, uint8_t* rgb, float* intensity, uint64_t* count) {
  return query_points_impl(ctx, shapes, shape_index, tree, &node, interval, capacity, mem, x, y, z, rgb, intensity, count);
}

// N4: the /nodes_data reply blob of octree_web_viewer (octree_web_viewer/src/backend.rs:90-177): per node
// min xyz (3 x f64 LE), edge (f64), num_points (u32), bytes per coordinate (u8), pad to 8, raw .xyz, pad to 8,
// raw .rgb, pad to 8. Returns the blob size in *needed; writes it when it fits in `capacity`.
extern "C" int pcv_octree_nodes_blob(pcv_octree* t, const uint64_t* node_indices, uint64_t count, uint8_t* out,
                                     uint64_t capacity, uint64_t* needed) {
  if (!t || !needed || (count && !node_indices)) return PCV_E_INVALID;
  auto pad8 = [](uint64_t v) { return (v + 7) & ~7ull; };
  uint64_t size = 0;
  for (uint64_t k = 0; k < count; ++k) {
    if (node_indices[k] >= t->nodes.size()) return t->ctx->fail(PCV_E_NOT_FOUND, "Could not get node.");
    const pcv_node_info& nd = t->nodes[node_indices[k]];
    const uint64_t np = (uint64_t)nd.num_points;
    size += pad8(32 + 4 + 1) + pad8(np * 3 * (uint64_t)pcv_bytes_per_coordinate(nd.encoding)) + pad8(np * 3);
  }
  *needed = size;
  if (!out || capacity < size) return PCV_OK;
  uint8_t* w = out;
  for (uint64_t k = 0; k < count; ++k) {
    const pcv_node_info& nd = t->nodes[node_indices[k]];
    const uint8_t *xyz, *rgbp;
    uint64_t lx, lr;
    int rc = pcv_octree_node_data(t, node_indices[k], 0, &xyz, &lx);
    if (rc) return rc;
    if ((rc = pcv_octree_node_data(t, node_indices[k], 1, &rgbp, &lr))) return rc;
    uint8_t* start = w;
    std::memcpy(w, nd.cube_min, 24);
    std::memcpy(w + 24, &nd.cube_edge, 8);
    const uint32_t np32 = (uint32_t)nd.num_points;
    std::memcpy(w + 32, &np32, 4);
    w[36] = (uint8_t)pcv_bytes_per_coordinate(nd.encoding);
    w += 37;
    while ((uint64_t)(w - start) % 8) *w++ = 0;
    std::memcpy(w, xyz, lx);
    w += lx;
    while ((uint64_t)(w - out) % 8) *w++ = 0;
    std::memcpy(w, rgbp, lr);
    w += lr;
    while ((uint64_t)(w - out) % 8) *w++ = 0;
  }
  return PCV_OK;
}
