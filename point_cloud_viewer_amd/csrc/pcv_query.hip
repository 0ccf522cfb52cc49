// pcv_query.hip — batched frustum / OBB / AABB transform-and-cull for gfx950.
//
//   K7a shape_setup      Frustum::from_matrix4 / intersector / cache_separating_axes_for_aabb
//                        (reference src/geometry/frustum.rs:111-166, src/math/sat.rs:111-143), Obb (obb.rs:48-80)
//   K7  cull_nodes       sat() of every (shape, node cube) pair (sat.rs:174-205) + relative_size_on_screen
//                        (src/octree/mod.rs:119-139)
//   K7b visible_nodes    Octree::get_visible_nodes — best-first traversal with Rust's BinaryHeap order
//                        (octree/mod.rs:228-283,360-404), one wave per frustum
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
  std::vector<int32_t> kinds;  // host copy: the point kernels are compiled per shape kind
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

// sat() of one cube against one prepared shape: Out if any axis separates, else Cross if B sticks out on any axis,
// else In (sat.rs:174-194) — so the walk over the axes stops at the first separating one, like the reference's early
// return, and the Relation does not depend on where it stops.
// The interval of the cube's 8 corners on an axis: each corner is fl(fl(x a_x + y a_y) + z a_z) with x, y, z the low or
// high bound; rounding is monotone, so the least (greatest) corner is the one built from the three least (greatest)
// products — 6 min/max + 4 adds instead of 16 adds + 16 min/max. Only when a bound comes out non-finite (inf / NaN
// inputs) are the 8 corners folded literally, in aabb.rs:114-125 order, so that f64::min / max skip NaNs as they do
// in the reference.
// the interval of the cube [l, h]^3 on one axis (see sat_cube): its three least / greatest products, or — non-finite bounds — its
// eight corners folded literally
__device__ __forceinline__ void sat_axis_interval(double lx, double hx, double ly, double hy, double lz, double hz, double ax, double ay,
                                                  double az, double& bmin, double& bmax, double& magnitude) {
  const double plx = lx * ax, phx = hx * ax, ply = ly * ay, phy = hy * ay, plz = lz * az, phz = hz * az;
  bmin = (fmin(plx, phx) + fmin(ply, phy)) + fmin(plz, phz);
  bmax = (fmax(plx, phx) + fmax(ply, phy)) + fmax(plz, phz);
  magnitude = ((fabs(plx) + fabs(phx)) + (fabs(ply) + fabs(phy))) + (fabs(plz) + fabs(phz));
  if (!(fabs(bmin) <= 1.7976931348623157e308 && fabs(bmax) <= 1.7976931348623157e308)) {
    // corners in aabb.rs:114-125 order: (l,l,l) (h,l,l) (l,h,l) (h,h,l) (l,l,h) (h,l,h) (l,h,h) (h,h,h)
    double c0 = (plx + ply) + plz, c1 = (phx + ply) + plz, c2 = (plx + phy) + plz, c3 = (phx + phy) + plz;
    double c4 = (plx + ply) + phz, c5 = (phx + ply) + phz, c6 = (plx + phy) + phz, c7 = (phx + phy) + phz;
    bmin = fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(1.7976931348623157e308, c0), c1), c2), c3), c4), c5), c6), c7);
    bmax = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(-1.7976931348623157e308, c0), c1), c2), c3), c4), c5), c6), c7);
  }
}
__device__ __forceinline__ int sat_cube(const PcvShapeDev* __restrict__ s, double mnx, double mny, double mnz, double edge) {
  if (s->kind == PCV_SHAPE_ALL) return 1;  // AllPoints intersects everything (math/mod.rs:139-160) -> "not Out"
  // Cube::to_aabb: Aabb::new(min, min + edge) (inf / sup)
  const double ax_ = mnx + edge, ay_ = mny + edge, az_ = mnz + edge;
  const double lx = fmin(mnx, ax_), hx = fmax(mnx, ax_);
  const double ly = fmin(mny, ay_), hy = fmax(mny, ay_);
  const double lz = fmin(mnz, az_), hz = fmax(mnz, az_);
  bool cross = false;
  const int na = s->naxes;
  for (int a = 0; a < na; ++a) {
    double bmin, bmax, mag;
    sat_axis_interval(lx, hx, ly, hy, lz, hz, s->axes[3 * a], s->axes[3 * a + 1], s->axes[3 * a + 2], bmin, bmax, mag);
    const double amin = s->amin[a], amax = s->amax[a];
    if (bmin > amax || bmax < amin) return 2;
    cross = cross || (amin > bmin || bmax > amax);
  }
  return cross ? 1 : 0;
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

// K7s (round 5): the same relations as a LIST per shape. 99.8 % of the (frustum, node) pairs of BASELINE config 4 are Out; the
// dense matrix spends most of its time on the size on screen of pairs nobody looks at (two IEEE divisions per corner) and
// its 546 MB on the way to the host. One workgroup per shape walks the node table in tiles of 256 and appends the nodes
// that are not Out IN NODE ORDER: {node index, relation, relative_size_on_screen} — the size is computed for those only,
// which is exactly where the reference computes it (octree/mod.rs:261-272: a node is projected when it is pushed).
__global__ __launch_bounds__(256) void cull_nodes_sparse_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t m,
                                                                 const double* __restrict__ cubes /* m x 4 */, uint32_t capacity,
                                                                 uint32_t* __restrict__ counts, uint32_t* __restrict__ out_node,
                                                                 uint8_t* __restrict__ out_rel, double* __restrict__ out_size,
                                                                 const uint32_t* __restrict__ redo /* set: only the flagged shapes */) {
  __shared__ uint32_t wave_tot[4];
  if (redo && !redo[blockIdx.x]) return;  // (uniform) the tree walk finished this shape
  const PcvShapeDev* s = shapes + blockIdx.x;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t row = (uint64_t)blockIdx.x * capacity;
  uint32_t base = 0;  // entries of this shape so far (uniform)
  const bool valid = s->valid != 0;
  for (uint32_t t0 = 0; t0 < m; t0 += 256) {
    const uint32_t i = t0 + threadIdx.x;
    double4 c = make_double4(0, 0, 0, 0);
    int rel = 2;
    if (i < m && valid) {
      c = *reinterpret_cast<const double4*>(cubes + 4 * (uint64_t)i);
      rel = sat_cube(s, c.x, c.y, c.z, c.w);
    }
    const bool keep = rel != 2;
    const uint64_t b = __ballot(keep);
    if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t v = wave_tot[w];
      before += w < wave ? v : 0u;
      total += v;
    }
    if (keep) {
      const uint32_t pos = base + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      if (pos < capacity) {
        out_node[row + pos] = i;
        out_rel[row + pos] = (uint8_t)rel;
        if (out_size) out_size[row + pos] = size_on_screen(s->clip_from_query, c.x, c.y, c.z, c.w);
      }
    }
    base += total;
    __syncthreads();  // wave_tot is rewritten by the next tile
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = base;
}

// K7t (round 6): the same lists, descending the tree like the reference's own traversals do (octree_iterator.rs:30-43,
// octree/mod.rs:261-272: children are only tested under a parent that is not Out). 99.76 % of the pairs of BASELINE config 4 are
// Out and nearly all of them sit under an Out ancestor. One WAVE per shape walks the tree breadth first — node order is
// (level, index), so the breadth-first order of the kept nodes IS the list's order.
//   * The wave's lanes are the shape's AXES, not the children: lane l holds axis l mod 32 of the shape (<= 26) in registers for
//     the whole walk, lanes 0-31 test one child of the popped node, lanes 32-63 the next, and three ballots give both Relations
//     (Out if any axis separates, else Cross if the cube sticks out on any axis, else In: sat.rs:174-194 does not depend on the
//     order of the axes). A first form with one child per lane and the loop over the axes inside ran 286-370 us for the 10 000
//     frusta: every round paid all 26 axes for a handful of busy lanes; the flat kernel needed 431.
//   * The queue (LDS) holds the kept INNER nodes only (87 % of a tree's nodes are leaves: listed, never expanded), each with its
//     cube, first child and child mask, so a popped node costs no dependent global load: its children's cubes are the recurrence
//     step NodeId::find_bounding_cube takes (node.rs:160-170: edge /= 2; min += bit * edge) — how the table's own cubes were
//     made (tests/test_gpu_query.py checks it on the node table) — and the children's own masks are requested (lanes 0-7) before
//     the tests and used after them.
//   * relative_size_on_screen of a popped node's kept children: eight lanes per child, one corner each (size_on_screen_by_corner).
// A subtree is skipped only under a node that is Out BY A MARGIN: some axis separates it by more than 1e-9 of the magnitudes
// involved — ~10^6 times the rounding error of any cube inside this one (a descendant's bounds lie within a few ulps of its
// ancestor's: min += bit * edge only adds, max = min + edge) — so every descendant is Out for the flat evaluation too. A shape
// that meets an Out node without that margin (a face within an ulp of a cube face, non-finite bounds), whose frontier outgrows
// the queue, or that is AllPoints, is flagged and redone by the flat kernel (cull_nodes_sparse_kernel with `redo`): the lists
// are the flat kernel's in every case.
__device__ __forceinline__ double size_on_screen_by_corner(const double* __restrict__ m, double mnx, double mny, double mnz, double edge,
                                                           uint32_t corner) {
  // lanes 8 g .. 8 g + 7 take the eight corners of cube g (size_on_screen's order), each its own projection and its two divisions;
  // the corners' clamped x / y are folded with min / max across the eight lanes (a min / max over a set: the order of the fold only
  // decides the sign of a zero)
  const double px = ((0x56u >> corner) & 1u) ? mnx + edge : mnx, py = ((0x9au >> corner) & 1u) ? mny + edge : mny,
               pz = ((0xe2u >> corner) & 1u) ? mnz + edge : mnz;
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = ((M4(m, r, 0) * px + M4(m, r, 1) * py) + M4(m, r, 2) * pz) + M4(m, r, 3) * 1.0;
  bool bad = v[3] == 0.0;
  double lox = clamp_num(v[0] / v[3], -1., 1.), loy = clamp_num(v[1] / v[3], -1., 1.);
  double hix = lox, hiy = loy;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    lox = fmin(lox, __shfl_xor(lox, o, 64));
    hix = fmax(hix, __shfl_xor(hix, o, 64));
    loy = fmin(loy, __shfl_xor(loy, o, 64));
    hiy = fmax(hiy, __shfl_xor(hiy, o, 64));
    bad = bad || (__shfl_xor((int)bad, o, 64) != 0);
  }
  if (bad) return __longlong_as_double(0x7ff8000000000000LL);
  return (hix - lox) * (hiy - loy);
}
struct CullEntry {
  double mnx, mny, mnz, edge;
  uint32_t first_child, mask;
};
constexpr uint32_t kCullQueue = 128;  // kept inner nodes waiting for their children to be tested, per wave (5 KiB: LDS does not bound the occupancy)
// this lane's axis against the cube (mn, mn + edge): does it separate (Out), does the cube stick out (Cross), does it separate by
// the margin
struct AxisTest {
  bool sep, cross, robust;
};
__device__ __forceinline__ AxisTest cull_axis_test(bool on, double ax, double ay, double az, double amin, double amax, double mnx,
                                                   double mny, double mnz, double edge) {
  const double ax_ = mnx + edge, ay_ = mny + edge, az_ = mnz + edge;  // Cube::to_aabb, as sat_cube
  const double lx = fmin(mnx, ax_), hx = fmax(mnx, ax_), ly = fmin(mny, ay_), hy = fmax(mny, ay_), lz = fmin(mnz, az_), hz = fmax(mnz, az_);
  double bmin, bmax, mag;
  sat_axis_interval(lx, hx, ly, hy, lz, hz, ax, ay, az, bmin, bmax, mag);
  AxisTest t;
  t.sep = on && (bmin > amax || bmax < amin);
  t.cross = on && (amin > bmin || bmax > amax);
  const double scale = mag + (fabs(amin) + fabs(amax));
  t.robust = on && fmax(bmin - amax, amin - bmax) > 1e-9 * scale && scale <= 1.7976931348623157e308;  // (NaN / inf anywhere: no)
  return t;
}
// HIER: the walk IS the answer — PointCloud::nodes_in_location (octree/mod.rs:309-323, NodeIdsIterator: a node's children are
// visited iff the node is not Out): every Out node prunes its subtree, margin or not; node indices only.
template <bool SIZES, bool HIER = false>
__global__ __launch_bounds__(256) void cull_nodes_tree_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t nshapes, uint32_t m,
                                                               const double* __restrict__ cubes /* m x 4, find_bounding_cube */,
                                                               const uint32_t* __restrict__ first_child, const uint8_t* __restrict__ child_mask,
                                                               uint32_t capacity, uint32_t* __restrict__ counts, uint32_t* __restrict__ out_node,
                                                               uint8_t* __restrict__ out_rel, double* __restrict__ out_size,
                                                               uint32_t* __restrict__ redo) {
  __shared__ CullEntry queue[4][kCullQueue];
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const uint32_t f = blockIdx.x * 4 + wave;
  if (f >= nshapes) return;  // wave-uniform
  const PcvShapeDev* s = shapes + f;
  const uint64_t row = (uint64_t)f * capacity;
  CullEntry* q = queue[wave];
  uint32_t head = 0, tail = 0;  // queue of kept inner nodes
  uint32_t nout = 0;            // listed nodes
  const int na = s->naxes;
  bool again = s->kind == PCV_SHAPE_ALL;  // (every node: the flat kernel lists them as fast)
  // this lane's axis, for the whole walk
  const uint32_t axis = lane & 31u;
  const bool on = (int)axis < na;
  double ax = 0, ay = 0, az = 0, amin = 0, amax = 0;
  if (on) {
    ax = s->axes[3 * axis], ay = s->axes[3 * axis + 1], az = s->axes[3 * axis + 2];
    amin = s->amin[axis], amax = s->amax[axis];
  }
  if (s->valid && !again) {
    {  // the root (both halves of the wave test it: the lower one's ballot bits are read)
      const double4 c = *reinterpret_cast<const double4*>(cubes);
      const AxisTest t = cull_axis_test(on, ax, ay, az, amin, amax, c.x, c.y, c.z, c.w);
      const uint32_t sep = (uint32_t)__ballot(t.sep), cross = (uint32_t)__ballot(t.cross), rob = (uint32_t)__ballot(t.robust);
      if (sep == 0u) {
        const uint32_t cm = child_mask[0];
        if (lane == 0) {
          if (capacity) {  // (entries past `capacity` are dropped, the count is not: capacity 0 only counts)
            out_node[row] = 0;
            if (!HIER) out_rel[row] = (uint8_t)(cross ? 1 : 0);
            if (SIZES) out_size[row] = size_on_screen(s->clip_from_query, c.x, c.y, c.z, c.w);
          }
          q[0] = CullEntry{c.x, c.y, c.z, c.w, first_child[0], cm};
        }
        nout = 1;
        tail = cm ? 1u : 0u;
      } else {
        again = !HIER && rob == 0u;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    while (head < tail && !again) {
      const CullEntry e = q[head & (kCullQueue - 1u)];  // (one address: a broadcast)
      const uint32_t pmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.mask);
      const uint32_t pfirst = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.first_child);
      // the children's own masks / first children: requested now (lanes 0-7), used after the tests
      uint32_t cm = 0, cf = 0;
      const uint32_t mychild = pfirst + (uint32_t)__popc(pmask & ((1u << (lane & 7u)) - 1u));
      if (lane < 8u && ((pmask >> lane) & 1u)) {
        cm = child_mask[mychild];
        cf = first_child[mychild];
      }
      const double half = e.edge / 2.0;  // node.rs:160-170
      uint32_t kept_mask = 0, cross_mask = 0;  // per digit (wave-uniform)
      for (uint32_t rest = pmask; rest != 0u && !again;) {
        const uint32_t d0 = (uint32_t)__builtin_ctz(rest);
        rest &= rest - 1u;
        const uint32_t d1 = rest ? (uint32_t)__builtin_ctz(rest) : 8u;
        rest &= rest - 1u;  // (0 & anything stays 0)
        const uint32_t digit = lane < 32u ? d0 : d1;
        const bool lane_on = on && digit < 8u;
        const double cx = e.mnx + ((digit & 4u) ? half : 0.0), cy = e.mny + ((digit & 2u) ? half : 0.0), cz = e.mnz + ((digit & 1u) ? half : 0.0);
        const AxisTest t = cull_axis_test(lane_on, ax, ay, az, amin, amax, cx, cy, cz, half);
        const uint64_t sep = __ballot(t.sep), cross = __ballot(t.cross), rob = __ballot(t.robust);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t d = h ? d1 : d0;
          if (d >= 8u) continue;
          const uint32_t sp = (uint32_t)(sep >> (32 * h)), cr = (uint32_t)(cross >> (32 * h)), rb = (uint32_t)(rob >> (32 * h));
          if (sp == 0u) {
            kept_mask |= 1u << d;
            if (cr) cross_mask |= 1u << d;
          } else if (!HIER && rb == 0u) {
            again = true;  // Out without the margin: its subtree cannot be skipped
          }
        }
      }
      const uint32_t kept = (uint32_t)__popc(kept_mask);
      if (again) break;
      const bool mine = lane < 8u && ((kept_mask >> lane) & 1u);
      const uint32_t k = (uint32_t)__popc(kept_mask & ((1u << (lane & 7u)) - 1u));
      if (mine && nout + k < capacity) {  // (entries past `capacity` are dropped, the count is not)
        out_node[row + nout + k] = mychild;
        if (!HIER) out_rel[row + nout + k] = (uint8_t)((cross_mask >> lane) & 1u);
      }
      if (SIZES && kept && nout < capacity) {  // lanes 8 g .. 8 g + 7: the eight corners of the g-th kept child
        const uint32_t g = lane >> 3;
        uint32_t mk = kept_mask;
        for (uint32_t i = 0; i < g && mk; ++i) mk &= mk - 1u;  // drop the g lowest set bits
        const uint32_t d = mk ? (uint32_t)__builtin_ctz(mk) : (uint32_t)__builtin_ctz(kept_mask);  // (idle groups redo the first: no divergence)
        const double cx = e.mnx + ((d & 4u) ? half : 0.0), cy = e.mny + ((d & 2u) ? half : 0.0), cz = e.mnz + ((d & 1u) ? half : 0.0);
        const double sz = size_on_screen_by_corner(s->clip_from_query, cx, cy, cz, half, lane & 7u);
        if (g < kept && nout + g < capacity && (lane & 7u) == 0u) out_size[row + nout + g] = sz;
      }
      const bool inner = mine && cm != 0u;
      const uint32_t inner_mask = (uint32_t)__ballot(inner);
      const uint32_t pushed = (uint32_t)__popc(inner_mask);
      if (tail + pushed - (head + 1u) > kCullQueue) {  // a frontier wider than the queue: the flat kernel
        again = true;
        break;
      }
      if (inner) {
        const double cx = e.mnx + ((lane & 4u) ? half : 0.0), cy = e.mny + ((lane & 2u) ? half : 0.0), cz = e.mnz + ((lane & 1u) ? half : 0.0);
        q[(tail + (uint32_t)__popc(inner_mask & ((1u << lane) - 1u))) & (kCullQueue - 1u)] = CullEntry{cx, cy, cz, half, cf, cm};
      }
      nout += kept;
      tail += pushed;
      head += 1u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (lane == 0) {
    counts[f] = nout;
    redo[f] = again ? 1u : 0u;
  }
}

struct QTree {
  uint32_t m;
  const double* cubes;         // get_child-style cubes (min xyz, edge), node order = (level, index)
  const uint32_t* first_child;
  const uint8_t* child_mask;
  const uint8_t* empty;        // num_points == 0
};

// K7b: one wave per frustum. Lanes 0..7 run the SAT + size_on_screen of the popped node's children side by side;
// lane 0 owns the BinaryHeap (std's pop / push sift order restated, so the pop order is the reference's). The first
// kHeapLds heap slots live in LDS, anything deeper in the frustum's global scratch (m entries).
// Round 6: an entry carries what popping it needs — the node's first child, its child mask, whether it holds points — fetched
// when the node was PUSHED (beside the SAT / size arithmetic of its siblings), so a pop is followed by ONE global load (the node's
// cube; its children's cubes are Node::get_child steps from it, node.rs:190-211) instead of two dependent rounds of them.
struct HeapEntry {
  double size;
  uint32_t node;
  uint32_t first_child;
  uint32_t bits;  // child mask in bits 0..7, bit 8: Relation::Cross (else In), bit 9: the node holds no points
  uint32_t pad;
};
constexpr uint32_t kHeapLds = 256;  // 24 B entries: 4 waves x 6 KiB per workgroup
struct WaveHeap {
  HeapEntry* lds;
  HeapEntry* glb;  // indexed by heap slot too (its first kHeapLds slots stay unused)
  __device__ __forceinline__ HeapEntry get(uint32_t i) const { return i < kHeapLds ? lds[i] : glb[i]; }
  __device__ __forceinline__ void set(uint32_t i, const HeapEntry& e) const {
    if (i < kHeapLds) lds[i] = e;
    else glb[i] = e;
  }
};
__device__ __forceinline__ void heap_sift_up(const WaveHeap& d, uint32_t start, uint32_t pos) {
  HeapEntry elt = d.get(pos);
  while (pos > start) {
    uint32_t parent = (pos - 1) / 2;
    HeapEntry pe = d.get(parent);
    if (elt.size <= pe.size) break;
    d.set(pos, pe);
    pos = parent;
  }
  d.set(pos, elt);
}
// BinaryHeap::pop: swap the last element in, sift_down_to_bottom, sift_up
__device__ __forceinline__ HeapEntry heap_pop(const WaveHeap& d, uint32_t& len) {
  HeapEntry item = d.get(len - 1);
  --len;
  if (len > 0) {
    HeapEntry top = d.get(0);
    d.set(0, item);
    item = top;
    const uint32_t end = len;
    uint32_t pos = 0, child = 1;
    HeapEntry elt = d.get(0);
    while (child + 1 < end) {
      HeapEntry l = d.get(child), r = d.get(child + 1);
      const bool right = l.size <= r.size;
      child += right ? 1u : 0u;
      d.set(pos, right ? r : l);
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) {
      d.set(pos, d.get(child));
      pos = child;
    }
    d.set(pos, elt);
    heap_sift_up(d, 0, pos);
  }
  return item;
}
__global__ __launch_bounds__(256) void visible_nodes_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t first_shape,
                                                             uint32_t nshapes, QTree t, HeapEntry* __restrict__ heaps,
                                                             uint32_t capacity, uint32_t* __restrict__ counts,
                                                             uint32_t* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ HeapEntry lds_heap[4][kHeapLds];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t li = blockIdx.x * 4 + wave;
  if (li >= nshapes) return;  // wave-uniform
  const uint32_t f = first_shape + li;
  const PcvShapeDev* s = shapes + f;
  const WaveHeap d{lds_heap[wave], heaps + (uint64_t)li * t.m};
  uint32_t* o = out + (uint64_t)f * capacity;
  uint32_t len = 0, nout = 0;  // lane 0's
  int32_t st = 0;
  // this lane's axis of the shape, for the whole traversal (lanes 0-31 and 32-63 hold the same axes)
  const bool on = (int)(lane & 31u) < s->naxes;
  double ax = 0, ay = 0, az = 0, amin = 0, amax = 0;
  if (on) {
    ax = s->axes[3 * (lane & 31u)], ay = s->axes[3 * (lane & 31u) + 1], az = s->axes[3 * (lane & 31u) + 2];
    amin = s->amin[lane & 31u], amax = s->amax[lane & 31u];
  }
  if (!s->valid) {  // .expect("Invalid projection matrix.")
    if (lane == 0) {
      counts[f] = 0;
      status[f] = 1;
    }
    return;
  }
  if (t.m > 0 && lane == 0) {  // maybe_push_node(root, Cross)
    double sz = size_on_screen(s->clip_from_query, t.cubes[0], t.cubes[1], t.cubes[2], t.cubes[3]);
    if (sz != sz) st = 2;
    d.set(0, HeapEntry{sz, 0u, t.first_child[0], (uint32_t)t.child_mask[0] | 0x100u | (t.empty[0] ? 0x200u : 0u), 0u});
    len = 1;
  }
  const bool all_points = s->kind == PCV_SHAPE_ALL;
  for (;;) {
    if (!__shfl((int)(len > 0 && st == 0), 0)) break;
    uint32_t node = 0, first = 0, bits = 0;
    if (lane == 0) {
      const HeapEntry item = heap_pop(d, len);
      node = item.node;
      first = item.first_child;
      bits = item.bits;
    }
    node = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);  // (lane 0 is the first active lane)
    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
    bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)bits);
    const uint32_t mask = bits & 0xffu;
    const bool cross_parent = (bits & 0x100u) != 0u;
    // the popped node's cube (one address for the wave) — the only load a pop waits for
    const double4 pc = *reinterpret_cast<const double4*>(t.cubes + 4 * (uint64_t)node);
    // what the children's own entries will need, requested now (lanes 0-7), used when they are pushed
    const uint32_t c = first + (uint32_t)__popc(mask & ((1u << (lane & 7)) - 1u));
    uint32_t cbits = 0, cfirst = 0;
    if (lane < 8 && ((mask >> lane) & 1u)) {
      cbits = (uint32_t)t.child_mask[c] | (t.empty[c] ? 0x200u : 0u);
      cfirst = t.first_child[c];
    }
    const double half = pc.w / 2.;  // Node::get_child (node.rs:190-211): min += half only where the bit is set
    // maybe_push_node on the children that exist: the wave's lanes are the shape's AXES — lanes 0-31 test one child, lanes 32-63
    // the next, two ballots give both Relations — and a kept child's size on screen is computed by eight lanes, one corner each
    // (one lane per child with the 26 axes and the 8 corners in loops left 56 lanes idle for ~3 000 instructions per pop)
    uint32_t kept_mask = mask, cross_mask = 0;  // children of an In node are In without a test (octree/mod.rs:261-272)
    if (cross_parent && all_points) {
      cross_mask = mask;  // sat_cube: AllPoints is "not Out" of everything, reported as Cross
    } else if (cross_parent) {
      kept_mask = 0;
      for (uint32_t rest = mask; rest != 0u;) {
        const uint32_t d0 = (uint32_t)__builtin_ctz(rest);
        rest &= rest - 1u;
        const uint32_t d1 = rest ? (uint32_t)__builtin_ctz(rest) : 8u;
        rest &= rest - 1u;
        const uint32_t digit = lane < 32u ? d0 : d1;
        const double cx = (digit & 4u) ? pc.x + half : pc.x, cy = (digit & 2u) ? pc.y + half : pc.y, cz = (digit & 1u) ? pc.z + half : pc.z;
        const AxisTest at = cull_axis_test(on && digit < 8u, ax, ay, az, amin, amax, cx, cy, cz, half);
        const uint64_t sep = __ballot(at.sep), cross = __ballot(at.cross);
        if ((uint32_t)sep == 0u) {
          kept_mask |= 1u << d0;
          if ((uint32_t)cross) cross_mask |= 1u << d0;
        }
        if (d1 < 8u && (uint32_t)(sep >> 32) == 0u) {
          kept_mask |= 1u << d1;
          if ((uint32_t)(cross >> 32)) cross_mask |= 1u << d1;
        }
      }
    }
    double sz = 0.0;  // lane 8 g: the size of the g-th kept child
    if (kept_mask) {
      const uint32_t g = lane >> 3;
      uint32_t mk = kept_mask;
      for (uint32_t i = 0; i < g && mk; ++i) mk &= mk - 1u;
      const uint32_t dg = (uint32_t)(mk ? __builtin_ctz(mk) : __builtin_ctz(kept_mask));  // (idle groups redo the first: no divergence)
      const double cx = (dg & 4u) ? pc.x + half : pc.x, cy = (dg & 2u) ? pc.y + half : pc.y, cz = (dg & 1u) ? pc.z + half : pc.z;
      sz = size_on_screen_by_corner(s->clip_from_query, cx, cy, cz, half, lane & 7u);
    }
    uint32_t g = 0;
    for (int ci = 0; ci < 8; ++ci) {  // pushes in child order, like the reference's loop
      if (!((kept_mask >> ci) & 1u)) continue;  // (wave-uniform)
      const double z = __shfl(sz, (int)(8u * g));
      const uint32_t cc = (uint32_t)__shfl((int)c, ci), cf = (uint32_t)__shfl((int)cfirst, ci), cb = (uint32_t)__shfl((int)cbits, ci);
      ++g;
      if (lane == 0) {
        if (z != z) st = 2;
        d.set(len, HeapEntry{z, cc, cf, cb | (((cross_mask >> ci) & 1u) << 8), 0u});
        heap_sift_up(d, 0, len);
        ++len;
      }
    }
    if (lane == 0 && !(bits & 0x200u)) {
      if (nout < capacity) o[nout] = node;
      ++nout;
    }
  }
  if (lane == 0) {
    counts[f] = nout;
    status[f] = st;
  }
}

// K7c: BFS of NodeIdsIterator; queue in global scratch.
__global__ __launch_bounds__(64) void nodes_in_location_kernel(const PcvShapeDev* __restrict__ shapes, uint32_t first_shape,
                                                                uint32_t nshapes, QTree t, const double* __restrict__ fb_cubes,
                                                                uint32_t* __restrict__ queues, uint32_t capacity,
                                                                uint32_t* __restrict__ counts, uint32_t* __restrict__ out,
                                                                const uint32_t* __restrict__ redo /* set: only the flagged shapes */) {
  const uint32_t li = blockIdx.x * 64 + threadIdx.x;
  if (li >= nshapes) return;
  const uint32_t f = first_shape + li;
  if (redo && !redo[f]) return;  // the wave-per-shape walk finished this one
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

// What contains() needs of a shape, fetched once per wave (wave-uniform: it lives in scalar registers) instead of
// once per point: the clip matrix (frustum), mins / maxs (AABB) or isometry + half extents (OBB). The point kernels
// are compiled per KIND (PCV_SHAPE_FRUSTUM stands for both frustum kinds), so the inner loops carry no shape switch.
template <int KIND>
struct ContainParams {
  double p[KIND == PCV_SHAPE_FRUSTUM ? 16 : KIND == PCV_SHAPE_OBB ? 10 : KIND == PCV_SHAPE_AABB ? 6 : 1];
};
template <int KIND>
__device__ __forceinline__ ContainParams<KIND> load_contain(const PcvShapeDev* __restrict__ shape) {
  ContainParams<KIND> c;
  if (KIND == PCV_SHAPE_AABB) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      c.p[i] = shape->bmin[i];
      c.p[3 + i] = shape->bmax[i];
    }
  } else if (KIND == PCV_SHAPE_FRUSTUM) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c.p[i] = shape->clip_from_query[i];
  } else if (KIND == PCV_SHAPE_OBB) {
#pragma unroll
    for (int i = 0; i < 7; ++i) c.p[i] = shape->iso[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.p[7 + i] = shape->half[i];
  } else {
    c.p[0] = 0.0;
  }
  return c;
}
template <int KIND>
__device__ __forceinline__ bool shape_contains(const ContainParams<KIND>& s, V3d p) {
  if (KIND == PCV_SHAPE_AABB) {  // aabb.rs:46-48: mins <= p < maxs
    return s.p[0] <= p.x && s.p[1] <= p.y && s.p[2] <= p.z && p.x < s.p[3] && p.y < s.p[4] && p.z < s.p[5];
  } else if (KIND == PCV_SHAPE_FRUSTUM) {  // frustum.rs:120-125
    const V3d c = m4_transform_point(s.p, p);
    const double mn = fmin(fmin(c.x, c.y), c.z), mx = fmax(fmax(c.x, c.y), c.z);
    return mn > -1.0 && mx < 1.0;
  } else if (KIND == PCV_SHAPE_OBB) {  // obb.rs:83-90
    const V3d q = v_add(quat_rotate(s.p + 3, p), V3d{s.p[0], s.p[1], s.p[2]});
    return fabs(q.x) <= s.p[7] && fabs(q.y) <= s.p[8] && fabs(q.z) <= s.p[9];
  }
  return true;  // AllPoints
}
// the kernel instance of a shape kind
#define PCV_DISPATCH_KIND(kind, CALL)                                  \
  switch (kind) {                                                      \
    case PCV_SHAPE_AABB: CALL(PCV_SHAPE_AABB); break;                  \
    case PCV_SHAPE_FRUSTUM:                                            \
    case PCV_SHAPE_FRUSTUM_WITH_INVERSE: CALL(PCV_SHAPE_FRUSTUM); break; \
    case PCV_SHAPE_OBB: CALL(PCV_SHAPE_OBB); break;                    \
    default: CALL(PCV_SHAPE_ALL); break;                               \
  }

// ---- wave chunks -------------------------------------------------------------------------------------------------
// A wave owns one chunk of consecutive points: kGroup x m of them, m chosen on the host so that the chunk's encoded
// bytes (3, 6, 12 or 24 per point; 16-byte aligned per node in the blob) fill at most kStageSlots 16-byte LDS slots.
// When the chunk lies in one node, those bytes are fetched with aligned 16-byte loads, all issued before the first is
// used (6 KiB in flight per wave — the kernel is latency bound, so bytes in flight are what buy bandwidth), and
// decoded from LDS; the keep flags leave as one dword store per lane and group of 256 points.
constexpr uint32_t kGroup = 256;                       // points per keep-store group (4 ballots)
constexpr uint32_t kStageSlots = kGroup * 24 / 16 + 1;  // uint4 slots per wave: 6 KiB of codes + the alignment skew
constexpr uint32_t kStageLoads = (kStageSlots + 63) / 64;

static inline uint32_t enc_stride_host(uint32_t enc) { return 3u * (uint32_t)pcv_bytes_per_coordinate(enc); }
// points per wave for nodes of at most `max_stride` encoded bytes per point
static inline uint32_t chunk_points(uint32_t max_stride) { return kGroup * (24u / max_stride); }

// The chunk's aligned 16-byte pieces, in registers between stage_issue (all loads in flight) and stage_commit (LDS).
// src = blob + off: the pointer keeps the blob's (global) address space, so the loads are global_load_dwordx4.
struct StageRegs {
  uint4 t0, t1, t2, t3, t4, t5, t6;
  uint32_t skew, n16;
};
static_assert(kStageLoads == 7, "the staging loads are written out by hand");
__device__ __forceinline__ StageRegs stage_issue(const uint8_t* __restrict__ blob, uint64_t off, uint32_t nbytes, uint32_t lane) {
  StageRegs r;
  r.skew = (uint32_t)((reinterpret_cast<uintptr_t>(blob) + off) & 15u);
  const uint4* g = reinterpret_cast<const uint4*>(blob + (off - r.skew));
  r.n16 = (r.skew + nbytes + 15u) >> 4;
  const uint32_t last = r.n16 - 1;  // slots past the end re-read the last one: no branches between the loads
  r.t0 = g[min(lane, last)];
  r.t1 = g[min(lane + 64u, last)];
  r.t2 = g[min(lane + 128u, last)];
  r.t3 = g[min(lane + 192u, last)];
  r.t4 = g[min(lane + 256u, last)];
  r.t5 = g[min(lane + 320u, last)];
  r.t6 = g[min(lane + 384u, last)];
  return r;
}
__device__ __forceinline__ void stage_commit(const StageRegs& r, uint4* stage, uint32_t lane) {
  if (lane < r.n16) stage[lane] = r.t0;
  if (lane + 64u < r.n16) stage[lane + 64u] = r.t1;
  if (lane + 128u < r.n16) stage[lane + 128u] = r.t2;
  if (lane + 192u < r.n16) stage[lane + 192u] = r.t3;
  if (lane + 256u < r.n16) stage[lane + 256u] = r.t4;
  if (lane + 320u < r.n16) stage[lane + 320u] = r.t5;
  if (lane + 384u < r.n16) stage[lane + 384u] = r.t6;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int ENC>
__device__ __forceinline__ V3d staged_point(const uint8_t* at, const double* cube_min, double cube_edge) {
  uint64_t c[3];
  if (ENC == PCV_ENC_UINT8) {
    c[0] = at[0];
    c[1] = at[1];
    c[2] = at[2];
  } else if (ENC == PCV_ENC_UINT16) {
    const uint16_t* q = reinterpret_cast<const uint16_t*>(at);
    c[0] = q[0];
    c[1] = q[1];
    c[2] = q[2];
  } else if (ENC == PCV_ENC_FLOAT32) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(at);
    c[0] = q[0];
    c[1] = q[1];
    c[2] = q[2];
  } else {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(at);
    c[0] = q[0];
    c[1] = q[1];
    c[2] = q[2];
  }
  return {pcv_decode_coord(ENC, c[0], cube_min[0], cube_edge), pcv_decode_coord(ENC, c[1], cube_min[1], cube_edge),
          pcv_decode_coord(ENC, c[2], cube_min[2], cube_edge)};
}

__device__ __forceinline__ uint32_t enc_stride(uint32_t enc) {
  return enc == PCV_ENC_UINT8 ? 3u : enc == PCV_ENC_UINT16 ? 6u : enc == PCV_ENC_FLOAT32 ? 12u : 24u;
}

// a group's keep flags from its 4 ballots (lane l of ballot r = point 64 r + l): lane l writes points 4 l .. 4 l + 3
__device__ __forceinline__ void store_keep(uint8_t* __restrict__ keep, uint32_t cnt, const unsigned long long b[4],
                                           uint32_t lane) {
  const uint32_t r = lane >> 4;
  const unsigned long long sel = r == 0 ? b[0] : r == 1 ? b[1] : r == 2 ? b[2] : b[3];
  const uint32_t bits = (uint32_t)(sel >> ((lane & 15u) * 4u)) & 0xfu;
  const uint32_t word = (bits * 0x00204081u) & 0x01010101u;  // bit j -> byte j
  const uint32_t q = lane * 4;
  if ((reinterpret_cast<uintptr_t>(keep) & 3u) == 0 && q + 3 < cnt) {
    *reinterpret_cast<uint32_t*>(keep + q) = word;
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
      if (q + j < cnt) keep[q + j] = (uint8_t)((word >> (8 * j)) & 1u);
  }
}

// keep flags of one chunk of `cnt` points of ONE node whose encoded bytes sit in the wave's LDS slice at `skew`
// (attr = the chunk's first attribute or null); returns how many were kept
template <int KIND, int ENC>
__device__ __forceinline__ uint32_t staged_keep_enc(const ContainParams<KIND>& shape, const uint4* stage, uint32_t skew,
                                                    const double* cube_min, double cube_edge, uint32_t cnt,
                                                    const float* __restrict__ attr, double lo, double hi, uint32_t lane,
                                                    uint8_t* __restrict__ keep) {
  constexpr uint32_t stride = ENC == PCV_ENC_UINT8 ? 3u : ENC == PCV_ENC_UINT16 ? 6u : ENC == PCV_ENC_FLOAT32 ? 12u : 24u;
  const uint8_t* bytes = reinterpret_cast<const uint8_t*>(stage) + skew;
  uint32_t tot = 0;
  for (uint32_t g0 = 0; g0 < cnt; g0 += kGroup) {
    unsigned long long b[4];
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      const uint32_t q = g0 + r * 64 + lane;
      bool k = false;
      if (q < cnt) {
        k = shape_contains<KIND>(shape, staged_point<ENC>(bytes + q * stride, cube_min, cube_edge));
        if (attr) {  // iterator.rs:82-91 + math/mod.rs:86-88
          const double a = (double)attr[q];
          k = k && (lo <= a && a <= hi);
        }
      }
      b[r] = __ballot(k);
    }
    store_keep(keep + g0, cnt - g0, b, lane);
    tot += (uint32_t)(__popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]));
  }
  return tot;
}
template <int KIND>
__device__ __forceinline__ uint32_t staged_keep(const ContainParams<KIND>& shape, const uint4* stage, uint32_t skew,
                                                uint32_t enc, const double* cube_min, double cube_edge, uint32_t cnt,
                                                const float* __restrict__ attr, double lo, double hi, uint32_t lane,
                                                uint8_t* __restrict__ keep) {
  switch (enc) {  // wave-uniform
    case PCV_ENC_UINT8:
      return staged_keep_enc<KIND, PCV_ENC_UINT8>(shape, stage, skew, cube_min, cube_edge, cnt, attr, lo, hi, lane, keep);
    case PCV_ENC_UINT16:
      return staged_keep_enc<KIND, PCV_ENC_UINT16>(shape, stage, skew, cube_min, cube_edge, cnt, attr, lo, hi, lane, keep);
    case PCV_ENC_FLOAT32:
      return staged_keep_enc<KIND, PCV_ENC_FLOAT32>(shape, stage, skew, cube_min, cube_edge, cnt, attr, lo, hi, lane, keep);
    default:
      return staged_keep_enc<KIND, PCV_ENC_FLOAT64>(shape, stage, skew, cube_min, cube_edge, cnt, attr, lo, hi, lane, keep);
  }
}

// K8 on one view (raw f64 SoA, or one node's encoded bytes): a wave per `chunk` points, one counter update per workgroup
template <int KIND>
__global__ __launch_bounds__(256) void cull_points_kernel(const PcvShapeDev* __restrict__ shape_dev, PointsView v, uint32_t chunk,
                                                           uint8_t* __restrict__ keep, unsigned long long* __restrict__ kept) {
  __shared__ uint4 stage[4][kStageSlots];
  __shared__ uint32_t wave_cnt[4];
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const uint64_t base = ((uint64_t)blockIdx.x * 4 + wave) * chunk;
  uint32_t tot = 0;
  if (base < v.n) {
    const ContainParams<KIND> shape = load_contain<KIND>(shape_dev);
    const uint32_t cnt = (uint32_t)(v.n - base < chunk ? v.n - base : chunk);
    if (v.encoded) {
      const StageRegs sr = stage_issue(v.encoded, base * enc_stride(v.enc), cnt * enc_stride(v.enc), lane);
      stage_commit(sr, stage[wave], lane);
      tot = staged_keep<KIND>(shape, stage[wave], sr.skew, v.enc, v.cube_min, v.cube_edge, cnt,
                        v.has_interval ? v.attr + base : nullptr, v.lo, v.hi, lane, keep + base);
    } else {
      for (uint32_t g0 = 0; g0 < cnt; g0 += kGroup) {
        unsigned long long b[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
          const uint32_t q = g0 + r * 64 + lane;
          bool k = false;
          if (q < cnt) {
            const uint64_t i = base + q;
            k = shape_contains<KIND>(shape, V3d{v.x[i], v.y[i], v.z[i]});
            if (v.has_interval) {
              const double a = (double)v.attr[i];
              k = k && (v.lo <= a && a <= v.hi);
            }
          }
          b[r] = __ballot(k);
        }
        store_keep(keep + base + g0, cnt - g0, b, lane);
        tot += (uint32_t)(__popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]));
      }
    }
  }
  if (lane == 0) wave_cnt[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (t) atomicAdd(kept, (unsigned long long)t);
  }
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
// One job per visited node, in traversal order. Every job is cut into chunks of its own size — as many points as
// fill a wave's LDS slice at the node's encoding (256 f64, 512 f32, 1024 u16, 2048 u8 points) — so a chunk never
// spans two nodes and the keep flags of a chunk start on a 4-byte boundary of the (internal) flag array.
struct QueryJob {
  uint64_t xyz_off;     // byte offset of the node's encoded positions in the xyz blob
  uint64_t point_off;   // point offset in the rgb / intensity blobs
  uint64_t keep_first;  // offset of the node's first flag (jobs are padded to 4 flags)
  uint32_t n;
  uint32_t enc;
  double cube_min[3];
  double cube_edge;
  uint32_t chunk_first;  // index of the node's first chunk
  uint32_t pad;
};

// pass 0 writes one descriptor per chunk, so that pass 1 has a single scalar load between "which chunk" and the
// staging loads
struct ChunkDesc {
  uint64_t src;         // offset of the chunk's first encoded byte in the xyz blob
  uint64_t attr_index;  // index of its first point in the rgb / intensity blobs
  double cube_min[3];
  double cube_edge;
  uint64_t keep_off;  // offset of its first flag
  uint32_t enc;
  uint32_t cnt;  // points
};
static_assert(sizeof(ChunkDesc) == 64, "one descriptor per s_load_dwordx16");

// the descriptor of chunk c, which belongs to job jb; chunks hold (LDS slice / stride) >> shift points
__host__ __device__ inline ChunkDesc make_chunk_desc(const QueryJob& jb, uint32_t c, uint32_t shift) {
  const uint32_t stride = jb.enc == PCV_ENC_UINT8 ? 3u : jb.enc == PCV_ENC_UINT16 ? 6u : jb.enc == PCV_ENC_FLOAT32 ? 12u : 24u;
  const uint32_t per = (kGroup * (24u / stride)) >> shift;
  const uint64_t kk = (uint64_t)(c - jb.chunk_first) * per;
  ChunkDesc d;
  d.src = jb.xyz_off + kk * stride;
  d.attr_index = jb.point_off + kk;
  d.cube_min[0] = jb.cube_min[0];
  d.cube_min[1] = jb.cube_min[1];
  d.cube_min[2] = jb.cube_min[2];
  d.cube_edge = jb.cube_edge;
  d.keep_off = jb.keep_first + kk;
  d.enc = jb.enc;
  d.cnt = (uint32_t)(jb.n - kk < per ? jb.n - kk : per);
  return d;
}

__global__ __launch_bounds__(256) void query_chunks_kernel(const QueryJob* __restrict__ jobs, uint32_t njobs, uint32_t nchunks,
                                                            uint32_t shift, ChunkDesc* __restrict__ desc) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  uint32_t lo = 0, hi = njobs;  // last job with chunk_first <= c
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobs[mid].chunk_first <= c) lo = mid;
    else hi = mid;
  }
  desc[c] = make_chunk_desc(jobs[lo], c, shift);
}

// pass 1: keep flag per point of every chunk + kept count per chunk. Persistent waves: wave w takes chunks w, w + W,
// ... and has the next chunk's encoded bytes in flight (registers) and the descriptor after that on its way while it
// decodes the current chunk from LDS, so a chunk's memory latency hides behind the f64 work of the one before.
template <int KIND>
__global__ __launch_bounds__(256) void query_flags_kernel(const PcvShapeDev* __restrict__ shape_dev,
                                                           const ChunkDesc* __restrict__ desc, uint32_t nchunks,
                                                           const uint8_t* __restrict__ xyz_blob,
                                                           const float* __restrict__ inten_blob, int has_interval, double lo,
                                                           double hi, uint8_t* __restrict__ keep,
                                                           uint32_t* __restrict__ chunk_counts) {
  __shared__ uint4 stage_all[4][kStageSlots];
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint4* stage = stage_all[wave];
  const uint32_t nwaves = gridDim.x * 4;
  uint32_t c = blockIdx.x * 4 + wave;
  if (c >= nchunks) return;
  const ContainParams<KIND> shape = load_contain<KIND>(shape_dev);
  ChunkDesc d = desc[c];  // wave-uniform: scalar loads
  ChunkDesc dn = desc[min(c + nwaves, nchunks - 1)];
  StageRegs sr = stage_issue(xyz_blob, d.src, d.cnt * enc_stride(d.enc), lane);
  for (;;) {
    const uint32_t skew = sr.skew;
    stage_commit(sr, stage, lane);
    const uint32_t cn = c + nwaves;
    if (cn < nchunks) sr = stage_issue(xyz_blob, dn.src, dn.cnt * enc_stride(dn.enc), lane);
    const ChunkDesc dnn = desc[min(cn + nwaves, nchunks - 1)];
    const uint32_t tot = staged_keep<KIND>(shape, stage, skew, d.enc, d.cube_min, d.cube_edge, d.cnt,
                                           has_interval ? inten_blob + d.attr_index : nullptr, lo, hi, lane, keep + d.keep_off);
    if (lane == 0) chunk_counts[c] = tot;
    if (cn >= nchunks) break;
    // the LDS slice is rewritten by the next commit: every lane must be done reading it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    c = cn;
    d = dn;
    dn = dnn;
  }
}

// exclusive scan of nb counters (16-byte aligned, allocated up to a multiple of 4) by one workgroup, in place; total to
// out_total.
// Up to 65 536 counters per pass sit in registers: thread t holds the uint4 of counters 4 (1024 j + t) .. + 3 for
// j < 16, all loads in flight at once; wave scans per j, one scan of the 256 (j, wave) totals, coalesced write-back.
__global__ __launch_bounds__(1024) void query_scan_kernel(uint32_t* __restrict__ counts, uint32_t nb,
                                                           unsigned long long* __restrict__ out_total) {
  constexpr int R = 16;
  __shared__ uint32_t tot[R * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* c4 = reinterpret_cast<uint4*>(counts);
  const uint32_t n4 = (nb + 3) / 4;
  unsigned long long running = 0;
  for (uint32_t base = 0; base < n4; base += R * 1024) {
    uint4 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t idx = base + j * 1024 + threadIdx.x;
      v[j] = idx < n4 ? c4[idx] : make_uint4(0, 0, 0, 0);
      if (4 * idx + 1 >= nb) v[j].y = 0;  // the allocation's tail past nb holds no counters
      if (4 * idx + 2 >= nb) v[j].z = 0;
      if (4 * idx + 3 >= nb) v[j].w = 0;
    }
    uint32_t inc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      uint32_t x = v[j].x + v[j].y + v[j].z + v[j].w;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(x, o, 64);
        if (lane >= o) x += t;
      }
      inc[j] = x;
      if (lane == 63) tot[j * 16 + wave] = x;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the 256 totals, 4 per lane, (j, wave) order
      uint32_t a0 = tot[4 * lane], a1 = tot[4 * lane + 1], a2 = tot[4 * lane + 2], a3 = tot[4 * lane + 3];
      uint32_t x = a0 + a1 + a2 + a3;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(x, o, 64);
        if (lane >= o) x += t;
      }
      uint32_t e = x - (a0 + a1 + a2 + a3);
      tot[4 * lane] = e;
      tot[4 * lane + 1] = e + a0;
      tot[4 * lane + 2] = e + a0 + a1;
      tot[4 * lane + 3] = e + a0 + a1 + a2;
    }
    __syncthreads();
    // positions fit u32 per octree (n < 2^32)
    uint32_t pass_total = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t idx = base + j * 1024 + threadIdx.x;
      const uint32_t s4 = v[j].x + v[j].y + v[j].z + v[j].w;
      const uint32_t e = (uint32_t)running + tot[j * 16 + wave] + inc[j] - s4;
      if (idx < n4) c4[idx] = make_uint4(e, e + v[j].x, e + v[j].x + v[j].y, e + v[j].x + v[j].y + v[j].z);
      if (j == R - 1 && threadIdx.x == 1023) pass_total = tot[j * 16 + wave] + inc[j];
    }
    // the pass total is known to the last thread only: broadcast it through LDS
    __syncthreads();
    if (threadIdx.x == 1023) tot[0] = pass_total;
    __syncthreads();
    running += tot[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_total = running;
}

// pass 2: stable compaction — decoded f64 positions, colours and intensity of the kept points, in job order; a wave
// per chunk, output position = the scanned chunk count + the rank among the chunk's kept points
__global__ __launch_bounds__(256) void query_compact_kernel(const ChunkDesc* __restrict__ desc, uint32_t nchunks,
                                                             const uint8_t* __restrict__ xyz_blob,
                                                             const uint8_t* __restrict__ rgb_blob,
                                                             const float* __restrict__ inten_blob,
                                                             const uint8_t* __restrict__ keep,
                                                             const uint32_t* __restrict__ chunk_offsets, uint64_t capacity,
                                                             double* __restrict__ ox, double* __restrict__ oy,
                                                             double* __restrict__ oz, uint8_t* __restrict__ orgb,
                                                             float* __restrict__ ointen) {
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const uint32_t ci = blockIdx.x * 4 + wave;
  if (ci >= nchunks) return;
  uint32_t pos0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_offsets[ci]);
  if (pos0 >= capacity) return;  // everything from here on is past the caller's buffers
  const ChunkDesc& d = desc[ci];
  PointsView v{};
  v.encoded = xyz_blob + d.src;
  v.enc = d.enc;
  v.cube_min[0] = d.cube_min[0];
  v.cube_min[1] = d.cube_min[1];
  v.cube_min[2] = d.cube_min[2];
  v.cube_edge = d.cube_edge;
  const uint8_t* kp = keep + d.keep_off;
  for (uint32_t q0 = 0; q0 < d.cnt; q0 += 64) {
    const uint32_t q = q0 + lane;
    const unsigned long long b = __ballot(q < d.cnt && kp[q]);
    const uint32_t pos = pos0 + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (((b >> lane) & 1ull) && pos < capacity) {
      const V3d p = load_point(v, q);
      ox[pos] = p.x;
      oy[pos] = p.y;
      oz[pos] = p.z;
      const uint8_t* c = rgb_blob + 3 * (d.attr_index + q);
      orgb[3 * (uint64_t)pos] = c[0];
      orgb[3 * (uint64_t)pos + 1] = c[1];
      orgb[3 * (uint64_t)pos + 2] = c[2];
      if (ointen) ointen[pos] = inten_blob[d.attr_index + q];
    }
    pos0 += (uint32_t)__popcll(b);
  }
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
  r->kinds.resize(count);
  for (uint32_t i = 0; i < count; ++i) r->kinds[i] = shapes[i].kind;
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

extern "C" int pcv_cull_nodes_sparse(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint32_t capacity, uint32_t* counts,
                                     uint32_t* node_indices, uint8_t* relation, double* size_on_screen_out) {
  if (!ctx) return PCV_E_INVALID;
  if (!shapes || !tree || !counts || (capacity && (!node_indices || !relation))) return ctx->fail(PCV_E_INVALID, "null argument");
  int rc = pcv_octree_prepare_query(tree);
  if (rc) return rc;
  const uint32_t m = tree->query->m, f = shapes->count;
  if (f == 0) return PCV_OK;
  if (m == 0) {
    std::memset(counts, 0, (size_t)f * 4);
    return PCV_OK;
  }
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  const size_t rows = (size_t)f * (capacity ? capacity : 1);
  uint32_t *d_cnt, *d_node;
  uint8_t* d_rel;
  double* d_sz = nullptr;
  if ((rc = sc.get(&d_cnt, f)) || (rc = sc.get(&d_node, rows)) || (rc = sc.get(&d_rel, rows))) return rc;
  if (size_on_screen_out && (rc = sc.get(&d_sz, rows))) return rc;
  uint32_t* d_redo;
  if ((rc = sc.get(&d_redo, f))) return rc;
  // PCV_CULL_FLAT=1 (libpcv_hip_exp.so): the flat kernel alone, as round 5 shipped it (the checker of the tree walk's lists)
  static const bool flat_only = [] {
    const char* e = pcv_experiment("PCV_CULL_FLAT");
    return e && atoi(e) != 0;
  }();
  {
    PcvProf prof(ctx, PCV_K_CULL_NODES_SPARSE);
    if (!flat_only) {
      if (d_sz)
        hipLaunchKernelGGL(cull_nodes_tree_kernel<true>, dim3((f + 3) / 4), dim3(256), 0, ctx->stream, shapes->dev, f, m, tree->query->fb_cubes,
                           tree->query->first_child, tree->query->child_mask, capacity, d_cnt, d_node, d_rel, d_sz, d_redo);
      else
        hipLaunchKernelGGL(cull_nodes_tree_kernel<false>, dim3((f + 3) / 4), dim3(256), 0, ctx->stream, shapes->dev, f, m, tree->query->fb_cubes,
                           tree->query->first_child, tree->query->child_mask, capacity, d_cnt, d_node, d_rel, d_sz, d_redo);
    }
    hipLaunchKernelGGL(cull_nodes_sparse_kernel, dim3(f), dim3(256), 0, ctx->stream, shapes->dev, m, tree->query->fb_cubes, capacity, d_cnt,
                       d_node, d_rel, d_sz, flat_only ? (const uint32_t*)nullptr : (const uint32_t*)d_redo);
  }
#ifdef PCV_EXPERIMENTS
  if (!flat_only && pcv_experiment("PCV_CULL_DEBUG")) {  // how many shapes the tree walk handed to the flat kernel
    std::vector<uint32_t> h(f);
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), d_redo, (size_t)f * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t r = 0;
    for (uint32_t v : h) r += v;
    fprintf(stderr, "[pcv cull] %u of %u shapes redone by the flat kernel\n", r, f);
  }
#endif
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(counts, d_cnt, (size_t)f * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (capacity) {
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(node_indices, d_node, rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(relation, d_rel, rows, hipMemcpyDeviceToHost, ctx->stream));
    if (d_sz) PCV_HIP_CHECK(ctx, hipMemcpyAsync(size_on_screen_out, d_sz, rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
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
  uint32_t* d_redo = nullptr;
  if (!visible && m > 0) {
    // round 6: one WAVE per shape walks the tree with the lanes as the shape's axes (cull_nodes_tree_kernel<.., HIER>): the
    // one-lane-per-shape walk below took 3.1 ms for 10 000 frusta; it stays for the shapes whose frontier outgrows the wave's
    // queue and for AllPoints
    if ((rc = sc.get(&d_redo, f))) return rc;
    PcvProf prof(ctx, PCV_K_NODES_IN_LOCATION);
    hipLaunchKernelGGL((cull_nodes_tree_kernel<false, true>), dim3((f + 3) / 4), dim3(256), 0, ctx->stream, shapes->dev, f, m, tree->query->fb_cubes,
                       tree->query->first_child, tree->query->child_mask, capacity, d_counts, d_out, (uint8_t*)nullptr, (double*)nullptr, d_redo);
  }
  for (uint32_t first = 0; first < f; first += batch) {
    const uint32_t nb = std::min(batch, f - first);
    PcvProf prof(ctx, visible ? PCV_K_VISIBLE_NODES : PCV_K_NODES_IN_LOCATION);
    if (visible)
      hipLaunchKernelGGL(visible_nodes_kernel, dim3((nb + 3) / 4), dim3(256), 0, ctx->stream, shapes->dev, first, nb, qt,
                         (HeapEntry*)scratch, capacity, d_counts, d_out, d_status);
    else
      hipLaunchKernelGGL(nodes_in_location_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, shapes->dev, first, nb,
                         qt, tree->query->fb_cubes, (uint32_t*)scratch, capacity, d_counts, d_out, (const uint32_t*)d_redo);
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
    const uint32_t chunk = v.encoded ? chunk_points(enc_stride_host(v.enc)) : kGroup;
#define PCV_CALL(K)                                                                                                     \
  hipLaunchKernelGGL(cull_points_kernel<K>, dim3((unsigned)((v.n + 4ull * chunk - 1) / (4ull * chunk))), dim3(256), 0, \
                     ctx->stream, shapes->dev + shape_index, v, chunk, d_keep, d_cnt)
    PCV_DISPATCH_KIND(shapes->kinds[shape_index], PCV_CALL)
#undef PCV_CALL
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
    const uint32_t chunk = v.encoded ? chunk_points(enc_stride_host(v.enc)) : kGroup;
#define PCV_CALL(K)                                                                                                     \
  hipLaunchKernelGGL(cull_points_kernel<K>, dim3((unsigned)((v.n + 4ull * chunk - 1) / (4ull * chunk))), dim3(256), 0, \
                     ctx->stream, shapes->dev + shape_index, v, chunk, d_keep, d_cnt)
    PCV_DISPATCH_KIND(shapes->kinds[shape_index], PCV_CALL)
#undef PCV_CALL
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
  // 2. one job per non-empty node, in traversal order. Small queries get smaller chunks (more waves, each with less
  //    serial work) and their descriptors straight from the host (no pass 0).
  uint64_t total = 0;
  for (uint32_t k = 0; k < nn; ++k) total += (uint64_t)std::max<int64_t>(tree->nodes[nodes[k]].num_points, 0);
  if (total == 0) return PCV_OK;
  const uint32_t shift = total < (1u << 19) ? 2u : total < (1u << 22) ? 1u : 0u;
  std::vector<QueryJob> jobs;
  uint64_t keep_total = 0;
  uint32_t nchunks = 0;
  for (uint32_t k = 0; k < nn; ++k) {
    const pcv_node_info& nd = tree->nodes[nodes[k]];
    if (nd.num_points <= 0) continue;
    QueryJob jb;
    jb.xyz_off = nd.xyz_offset;
    jb.point_off = nd.point_offset;
    jb.keep_first = keep_total;
    jb.n = (uint32_t)nd.num_points;
    jb.enc = nd.encoding;
    for (int a = 0; a < 3; ++a) jb.cube_min[a] = nd.cube_min[a];
    jb.cube_edge = nd.cube_edge;
    jb.chunk_first = nchunks;
    jb.pad = 0;
    jobs.push_back(jb);
    const uint32_t per = chunk_points(enc_stride_host(nd.encoding)) >> shift;
    nchunks += (uint32_t)(((uint64_t)nd.num_points + per - 1) / per);
    keep_total += ((uint64_t)nd.num_points + 3) & ~3ull;
  }
  const uint32_t njobs = (uint32_t)jobs.size();
  const bool host_desc = nchunks <= 2048;
  std::vector<ChunkDesc> h_desc;
  if (host_desc) {
    h_desc.reserve(nchunks);
    for (const QueryJob& jb : jobs) {
      const uint32_t per = chunk_points(enc_stride_host(jb.enc)) >> shift;
      const uint32_t nc = (jb.n + per - 1) / per;
      for (uint32_t c = 0; c < nc; ++c) h_desc.push_back(make_chunk_desc(jb, jb.chunk_first + c, shift));
    }
  }
  const uint32_t nb = (nchunks + 3) / 4;
  int cus = 0, per_cu = 0;
  PCV_HIP_CHECK(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
  // persistent pass 1: as many workgroups as are resident at once
#define PCV_CALL(K) PCV_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, query_flags_kernel<K>, 256, 0))
  PCV_DISPATCH_KIND(shapes->kinds[shape_index], PCV_CALL)
#undef PCV_CALL
  const uint32_t nb_flags = std::min<uint32_t>(nb, (uint32_t)std::max(cus, 1) * (uint32_t)std::max(per_cu, 1));
  QueryJob* d_jobs = nullptr;
  ChunkDesc* d_desc;
  uint8_t* d_keep;
  uint32_t* d_cc;
  unsigned long long* d_total;
  if ((!host_desc && (rc = sc.get(&d_jobs, njobs))) || (rc = sc.get(&d_keep, keep_total)) ||
      (rc = sc.get(&d_cc, ((size_t)nchunks + 3) & ~(size_t)3)) || (rc = sc.get(&d_total, 1)) || (rc = sc.get(&d_desc, nchunks)))
    return rc;
  if (host_desc)
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_desc, h_desc.data(), sizeof(ChunkDesc) * nchunks, hipMemcpyHostToDevice, ctx->stream));
  else
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(QueryJob) * njobs, hipMemcpyHostToDevice, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_CULL_POINTS);
    if (!host_desc)
      hipLaunchKernelGGL(query_chunks_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, ctx->stream, d_jobs, njobs, nchunks, shift,
                         d_desc);
#define PCV_CALL(K)                                                                                                      \
  hipLaunchKernelGGL(query_flags_kernel<K>, dim3(nb_flags), dim3(256), 0, ctx->stream, shapes->dev + shape_index, d_desc, \
                     nchunks, tree->d_xyz, (const float*)tree->d_int, interval ? 1 : 0, interval ? interval[0] : 0.0,      \
                     interval ? interval[1] : 0.0, d_keep, d_cc)
    PCV_DISPATCH_KIND(shapes->kinds[shape_index], PCV_CALL)
#undef PCV_CALL
  }
  hipLaunchKernelGGL(query_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_cc, nchunks, d_total);
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
      hipLaunchKernelGGL(query_compact_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_desc, nchunks, tree->d_xyz, tree->d_rgb,
                         (const float*)tree->d_int, d_keep, d_cc, nout, dx, dy, dz, drgb, want_int ? dint : nullptr);
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
