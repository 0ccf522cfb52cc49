// pcv_oracle_query.cpp — CPU ORACLE for the frustum / OBB / AABB transform-and-cull path (test infrastructure).
//
// Restates (paths relative to /root/reference):
//   src/math/sat.rs:67-205            Intersector, cache_separating_axes, sat, project_on_axis
//   src/geometry/frustum.rs:16-166    Perspective, Frustum::new/from_matrix4/contains/compute_corners/intersector
//   src/geometry/obb.rs:13-90         Obb::new/compute_corners/intersector/contains
//   src/geometry/aabb.rs:46-48,95-139 Aabb::contains/compute_corners/intersector, Cube::to_aabb
//   src/octree/mod.rs:103-139,228-283,309-323,360-404  project, relative_size_on_screen, get_visible_nodes,
//                                                       nodes_in_location_impl, OpenNode ordering, maybe_push_node
//   src/octree/octree_iterator.rs     NodeIdsIterator (BFS)
//   src/iterator.rs:82-119            FilteredIterator (per-point keep mask)
//
// Third-party arithmetic NOT under /root/reference, restated from the published algorithms of the pinned crates
// (Cargo.lock): nalgebra 0.22.0 — Matrix4::try_inverse (MESA-style cofactor 4x4 inverse, `1/det` then scale),
// Matrix4::transform_point (gemv column accumulation, divide by the normaliser unless zero), Unit::new_normalize
// (divide by sqrt of the left-to-right dot), Vector3::cross, UnitQuaternion * Vector3 (t = 2 q x v; v + w t + q x t),
// UnitQuaternion::to_rotation_matrix, Isometry3::inverse; Rust std BinaryHeap push/pop (sift_up /
// sift_down_to_bottom). These are pinned by the reference's tests only at the level of Relation / bool results
// (sat.rs:214-268, obb.rs:100-141, math/mod.rs:191-220, frustum.rs:178-205): ULP-level summation order is
// "parity unpinned"  and is stated here as the rule both this oracle and the HIP kernels follow.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <vector>

#include "pcv_oracle_core.h"

namespace pcvq {

using pcvo::Cube;
using pcvo::NodeId;
using pcvo::u128;

struct V3 {
  double x, y, z;
};
static inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // nalgebra dot, len 3
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 normalize(V3 v) {  // Unit::new_normalize: v / v.norm()
  double n = std::sqrt(dot(v, v));
  return {v.x / n, v.y / n, v.z / n};
}

// Column-major 4x4 (nalgebra storage): element (r, c) = m[c * 4 + r].
static inline double& at(double* m, int r, int c) { return m[c * 4 + r]; }
static inline double at(const double* m, int r, int c) { return m[c * 4 + r]; }

// out = a * b, gemv per column: out[i][j] = ((a[i][0] b[0][j] + a[i][1] b[1][j]) + a[i][2] b[2][j]) + a[i][3] b[3][j]
static void mat4_mul(const double* a, const double* b, double* out) {
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      double acc = at(a, i, 0) * at(b, 0, j);
      for (int k = 1; k < 4; ++k) acc = acc + at(a, i, k) * at(b, k, j);
      at(out, i, j) = acc;
    }
}

// nalgebra 0.22 linalg/inverse.rs do_inverse4 (cofactor expansion on the column-major slice).
static bool try_inverse4(const double* m, double* out) {
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

// nalgebra Matrix4::transform_point: (M3 p + t) / n with n = row3 . p + m33, unless n == 0.
static V3 transform_point(const double* m, V3 p) {
  double px[3] = {p.x, p.y, p.z};
  double r[3];
  for (int i = 0; i < 3; ++i) {
    double acc = at(m, i, 0) * px[0];
    acc = acc + at(m, i, 1) * px[1];
    acc = acc + at(m, i, 2) * px[2];
    r[i] = acc + at(m, i, 3);
  }
  double n = ((at(m, 3, 0) * px[0] + at(m, 3, 1) * px[1]) + at(m, 3, 2) * px[2]) + at(m, 3, 3);
  if (n != 0.0) return {r[0] / n, r[1] / n, r[2] / n};
  return {r[0], r[1], r[2]};
}

// Isometry3 = translation + unit quaternion (i, j, k, w).
struct Iso {
  V3 t;
  double q[4];
};
static V3 quat_rotate(const double q[4], V3 v) {  // UnitQuaternion * Vector3
  V3 qv = {q[0], q[1], q[2]};
  V3 t = scale(cross(qv, v), 2.0);
  V3 c = cross(qv, t);
  return add(add(scale(t, q[3]), c), v);
}
static V3 iso_transform_point(const Iso& s, V3 p) { return add(quat_rotate(s.q, p), s.t); }
static Iso iso_inverse(const Iso& s) {  // rotation.inverse() = conjugate; t' = R^-1 * (-t)
  Iso r;
  r.q[0] = -s.q[0];
  r.q[1] = -s.q[1];
  r.q[2] = -s.q[2];
  r.q[3] = s.q[3];
  r.t = quat_rotate(r.q, V3{-s.t.x, -s.t.y, -s.t.z});
  return r;
}
static void iso_to_homogeneous(const Iso& s, double* m) {  // UnitQuaternion::to_rotation_matrix + translation
  double i = s.q[0], j = s.q[1], k = s.q[2], w = s.q[3];
  double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0, ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
  std::memset(m, 0, 16 * sizeof(double));
  at(m, 0, 0) = ww + ii - jj - kk;
  at(m, 0, 1) = ij - wk;
  at(m, 0, 2) = wj + ik;
  at(m, 1, 0) = wk + ij;
  at(m, 1, 1) = ww - ii + jj - kk;
  at(m, 1, 2) = jk - wi;
  at(m, 2, 0) = ik - wj;
  at(m, 2, 1) = wi + jk;
  at(m, 2, 2) = ww - ii - jj + kk;
  at(m, 0, 3) = s.t.x;
  at(m, 1, 3) = s.t.y;
  at(m, 2, 3) = s.t.z;
  at(m, 3, 3) = 1.0;
}

// frustum.rs:16-80 Perspective::new / inverse
static void perspective_new(double left, double right, double bottom, double top, double near, double far, double* m) {
  std::memset(m, 0, 16 * sizeof(double));
  at(m, 0, 0) = (2.0 * near) / (right - left);
  at(m, 0, 2) = (right + left) / (right - left);
  at(m, 1, 1) = (2.0 * near) / (top - bottom);
  at(m, 1, 2) = (top + bottom) / (top - bottom);
  at(m, 2, 2) = -(far + near) / (far - near);
  at(m, 2, 3) = -(2.0 * far * near) / (far - near);
  at(m, 3, 2) = -1.0;
}
static void perspective_inverse(const double* p, double* m) {
  std::memset(m, 0, 16 * sizeof(double));
  at(m, 0, 0) = 1.0 / at(p, 0, 0);
  at(m, 0, 3) = at(p, 0, 2) / at(p, 0, 0);
  at(m, 1, 1) = 1.0 / at(p, 1, 1);
  at(m, 1, 3) = at(p, 1, 2) / at(p, 1, 1);
  at(m, 2, 3) = -1.0;
  at(m, 3, 2) = 1.0 / at(p, 2, 3);
  at(m, 3, 3) = at(p, 2, 2) / at(p, 2, 3);
}

// sat.rs:67-76 Intersector
struct Intersector {
  V3 corners[8];
  V3 edges[12];
  int ne = 0;
  V3 normals[6];
  int nn = 0;
};
struct Cached {  // sat.rs:155-158
  V3 corners[8];
  std::vector<V3> axes;
};

// frustum.rs:127-166
static Intersector frustum_intersector(const double* query_from_clip) {
  Intersector s;
  const double sg[2] = {-1.0, 1.0};
  int c = 0;
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz) s.corners[c++] = transform_point(query_from_clip, V3{sg[ix], sg[iy], sg[iz]});
  const V3* k = s.corners;
  s.edges[0] = normalize(sub(k[4], k[0]));
  s.edges[1] = normalize(sub(k[2], k[0]));
  s.edges[2] = normalize(sub(k[1], k[0]));
  s.edges[3] = normalize(sub(k[3], k[2]));
  s.edges[4] = normalize(sub(k[5], k[4]));
  s.edges[5] = normalize(sub(k[7], k[6]));
  s.ne = 6;
  s.normals[0] = normalize(cross(s.edges[0], s.edges[1]));
  s.normals[1] = normalize(cross(s.edges[0], s.edges[2]));
  s.normals[2] = normalize(cross(s.edges[0], s.edges[3]));
  s.normals[3] = normalize(cross(s.edges[1], s.edges[2]));
  s.normals[4] = normalize(cross(s.edges[1], s.edges[4]));
  s.nn = 5;
  return s;
}

// obb.rs:48-80
static Intersector obb_intersector(const Iso& query_from_obb, V3 h) {
  Intersector s;
  const double sx[8] = {-1, 1, -1, 1, -1, 1, -1, 1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
  for (int c = 0; c < 8; ++c) s.corners[c] = iso_transform_point(query_from_obb, V3{sx[c] * h.x, sy[c] * h.y, sz[c] * h.z});
  s.edges[0] = normalize(quat_rotate(query_from_obb.q, V3{1, 0, 0}));
  s.edges[1] = normalize(quat_rotate(query_from_obb.q, V3{0, 1, 0}));
  s.edges[2] = normalize(quat_rotate(query_from_obb.q, V3{0, 0, 1}));
  s.ne = 3;
  for (int i = 0; i < 3; ++i) s.normals[i] = s.edges[i];
  s.nn = 3;
  return s;
}

// aabb.rs:114-125 compute_corners
static void aabb_corners(const double mn[3], const double mx[3], V3 out[8]) {
  out[0] = {mn[0], mn[1], mn[2]};
  out[1] = {mx[0], mn[1], mn[2]};
  out[2] = {mn[0], mx[1], mn[2]};
  out[3] = {mx[0], mx[1], mn[2]};
  out[4] = {mn[0], mn[1], mx[2]};
  out[5] = {mx[0], mn[1], mx[2]};
  out[6] = {mn[0], mx[1], mx[2]};
  out[7] = {mx[0], mx[1], mx[2]};
}
// aabb.rs:127-139
static Intersector aabb_intersector(const double mn[3], const double mx[3]) {
  Intersector s;
  aabb_corners(mn, mx, s.corners);
  s.edges[0] = {1, 0, 0};
  s.edges[1] = {0, 1, 0};
  s.edges[2] = {0, 0, 1};
  s.ne = 3;
  for (int i = 0; i < 3; ++i) s.normals[i] = s.edges[i];
  s.nn = 3;
  return s;
}

// sat.rs:80-134 separating_axes_iter + cache_separating_axes (O(n^2) dedup with f64::EPSILON)
static Cached cache_separating_axes(const Intersector& s, const V3* other_edges, int noe, const V3* other_normals, int non) {
  std::vector<V3> all;
  for (int i = 0; i < s.nn; ++i) all.push_back(s.normals[i]);
  for (int i = 0; i < non; ++i) all.push_back(other_normals[i]);
  for (int i = 0; i < s.ne; ++i)
    for (int j = 0; j < noe; ++j) {
      V3 c = normalize(cross(s.edges[i], other_edges[j]));
      if (std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z)) all.push_back(c);
    }
  Cached out;
  for (int i = 0; i < 8; ++i) out.corners[i] = s.corners[i];
  for (const V3& a1 : all) {
    bool dupe = false;
    for (const V3& a2 : out.axes) {
      V3 dm = sub(a1, a2), dp = add(a1, a2);
      double d1 = dot(dm, dm), d2 = dot(dp, dp);
      if (std::fmin(d1, d2) < 2.220446049250313e-16) {
        dupe = true;
        break;
      }
    }
    if (!dupe) out.axes.push_back(a1);
  }
  return out;
}
static Cached cache_for_aabb(const Intersector& s) {  // sat.rs:138-143
  V3 unit[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  return cache_separating_axes(s, unit, 3, unit, 3);
}

enum { REL_IN = 0, REL_CROSS = 1, REL_OUT = 2 };

// sat.rs:196-205
static void project_on_axis(const V3* corners, int n, V3 axis, double* mn, double* mx) {
  double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
  for (int i = 0; i < n; ++i) {
    double p = dot(corners[i], axis);
    lo = std::fmin(lo, p);
    hi = std::fmax(hi, p);
  }
  *mn = lo;
  *mx = hi;
}
// sat.rs:174-194
static int sat(const V3* axes, int naxes, const V3* a, int na, const V3* b, int nb) {
  int rel = REL_IN;
  for (int i = 0; i < naxes; ++i) {
    double amin, amax, bmin, bmax;
    project_on_axis(a, na, axes[i], &amin, &amax);
    project_on_axis(b, nb, axes[i], &bmin, &bmax);
    if (bmin > amax || bmax < amin) return REL_OUT;
    if (amin > bmin || bmax > amax) rel = REL_CROSS;
  }
  return rel;
}
// sat.rs:146-152 Intersector::intersect (axes of both, not cached / not deduplicated)
static int intersect(const Intersector& a, const Intersector& b) {
  std::vector<V3> axes;
  for (int i = 0; i < a.nn; ++i) axes.push_back(a.normals[i]);
  for (int i = 0; i < b.nn; ++i) axes.push_back(b.normals[i]);
  for (int i = 0; i < a.ne; ++i)
    for (int j = 0; j < b.ne; ++j) {
      V3 c = normalize(cross(a.edges[i], b.edges[j]));
      if (std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z)) axes.push_back(c);
    }
  return sat(axes.data(), (int)axes.size(), a.corners, 8, b.corners, 8);
}

// Cube::to_aabb (aabb.rs:159-161,175-181) then compute_corners
static void cube_corners(const double mn[3], double edge, V3 out[8]) {
  double mx[3] = {mn[0] + edge, mn[1] + edge, mn[2] + edge};
  double lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {  // Aabb::new: inf / sup
    lo[a] = std::fmin(mn[a], mx[a]);
    hi[a] = std::fmax(mn[a], mx[a]);
  }
  aabb_corners(lo, hi, out);
}

// octree/mod.rs:103-139
static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static bool project(const double* m, V3 p, V3* out) {  // m * p.to_homogeneous(), from_homogeneous
  double v[4];
  for (int i = 0; i < 4; ++i) {
    double acc = at(m, i, 0) * p.x;
    acc = acc + at(m, i, 1) * p.y;
    acc = acc + at(m, i, 2) * p.z;
    v[i] = acc + at(m, i, 3) * 1.0;
  }
  if (v[3] == 0.0) return false;  // .unwrap() panics in the reference
  *out = {v[0] / v[3], v[1] / v[3], v[2] / v[3]};
  return true;
}
static double relative_size_on_screen(const double mn[3], double edge, const double* m, bool* ok) {
  double mx[3] = {mn[0] + edge, mn[1] + edge, mn[2] + edge};
  V3 pts[8] = {{mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]}, {mx[0], mn[1], mn[2]}, {mn[0], mx[1], mn[2]},
               {mx[0], mx[1], mn[2]}, {mn[0], mn[1], mx[2]}, {mx[0], mn[1], mx[2]}, {mn[0], mx[1], mx[2]}};
  double lo[2], hi[2];
  *ok = true;
  for (int i = 0; i < 8; ++i) {
    V3 q;
    if (!project(m, pts[i], &q)) {
      *ok = false;
      return 0;
    }
    double cx = clampd(q.x, -1., 1.), cy = clampd(q.y, -1., 1.);
    if (i == 0) {
      lo[0] = hi[0] = cx;
      lo[1] = hi[1] = cy;
    } else if (i == 1) {  // Aabb::new(a, b): inf / sup of the first two
      lo[0] = std::fmin(lo[0], cx);
      hi[0] = std::fmax(hi[0], cx);
      lo[1] = std::fmin(lo[1], cy);
      hi[1] = std::fmax(hi[1], cy);
    } else {  // grow
      lo[0] = std::fmin(lo[0], cx);
      hi[0] = std::fmax(hi[0], cx);
      lo[1] = std::fmin(lo[1], cy);
      hi[1] = std::fmax(hi[1], cy);
    }
  }
  return (hi[0] - lo[0]) * (hi[1] - lo[1]);
}

// A loaded octree for queries: id -> (num_points, cube) (octree/mod.rs:196-209)
struct QNode {
  int64_t num_points;
  Cube cube;
};
struct QOctree {
  Cube root_cube;
  std::map<u128, QNode> nodes;
};

// Rust std BinaryHeap<OpenNode> ordered by size_on_screen (octree/mod.rs:360-385).
struct OpenNode {
  NodeId id;
  Cube cube;
  int relation;
  double size;
  bool empty;
};
struct Heap {
  std::vector<OpenNode> d;
  void sift_up(size_t start, size_t pos) {
    OpenNode elt = d[pos];
    while (pos > start) {
      size_t parent = (pos - 1) / 2;
      if (elt.size <= d[parent].size) break;
      d[pos] = d[parent];
      pos = parent;
    }
    d[pos] = elt;
  }
  void push(const OpenNode& n) {
    size_t old = d.size();
    d.push_back(n);
    sift_up(0, old);
  }
  bool pop(OpenNode* out) {
    if (d.empty()) return false;
    OpenNode item = d.back();
    d.pop_back();
    if (!d.empty()) {
      std::swap(item, d[0]);
      // sift_down_to_bottom(0)
      size_t end = d.size(), pos = 0;
      OpenNode elt = d[0];
      size_t child = 1;
      while (child + 1 < end) {  // child <= end - 2
        child += (d[child].size <= d[child + 1].size) ? 1 : 0;
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
      }
      if (child == end - 1) {
        d[pos] = d[child];
        pos = child;
      }
      d[pos] = elt;
      sift_up(0, pos);
    }
    *out = item;
    return true;
  }
};

static Cube child_cube(const Cube& c, int ci) {  // Node::get_child node.rs:190-211
  Cube r;
  double half = c.edge / 2.;
  r.mn[0] = c.mn[0];
  r.mn[1] = c.mn[1];
  r.mn[2] = c.mn[2];
  if (ci & 1) r.mn[2] += half;
  if (ci & 2) r.mn[1] += half;
  if (ci & 4) r.mn[0] += half;
  r.edge = half;
  return r;
}

// octree/mod.rs:228-283 get_visible_nodes. Returns false where the reference would panic.
static bool get_visible_nodes(const QOctree& t, const double* matrix, std::vector<NodeId>* visible) {
  double inv[16];
  if (!try_inverse4(matrix, inv)) return false;  // .expect("Invalid projection matrix.")
  Cached isec = cache_for_aabb(frustum_intersector(inv));
  Heap open;
  bool ok = true;
  auto maybe_push = [&](int relation, NodeId id, const Cube& cube) {
    auto it = t.nodes.find(id.v);
    if (it == t.nodes.end()) return;
    bool good;
    double size = relative_size_on_screen(cube.mn, cube.edge, matrix, &good);
    if (!good) ok = false;
    open.push(OpenNode{id, cube, relation, size, it->second.num_points == 0});
  };
  maybe_push(REL_CROSS, NodeId::root(), t.root_cube);
  OpenNode cur;
  while (ok && open.pop(&cur)) {
    for (int ci = 0; ci < 8; ++ci) {
      Cube cc = child_cube(cur.cube, ci);
      NodeId cid = cur.id.get_child_id((uint8_t)ci);
      if (cur.relation == REL_CROSS) {
        V3 corners[8];
        cube_corners(cc.mn, cc.edge, corners);
        int rel = sat(isec.axes.data(), (int)isec.axes.size(), isec.corners, 8, corners, 8);
        if (rel == REL_OUT) continue;
        maybe_push(rel, cid, cc);
      } else {
        maybe_push(REL_IN, cid, cc);
      }
    }
    if (!cur.empty) visible->push_back(cur.id);
  }
  return ok;
}

// octree/mod.rs:309-323 + octree_iterator.rs: BFS, node passes iff its cube is not Out.
static void nodes_in_location(const QOctree& t, const Cached& isec, std::vector<NodeId>* out) {
  std::deque<NodeId> q;
  q.push_back(NodeId::root());
  while (!q.empty()) {
    NodeId cur = q.front();
    q.pop_front();
    auto it = t.nodes.find(cur.v);
    if (it == t.nodes.end()) continue;  // reference indexes `octree.nodes[&node_id]` (root must exist)
    V3 corners[8];
    cube_corners(it->second.cube.mn, it->second.cube.edge, corners);
    if (sat(isec.axes.data(), (int)isec.axes.size(), isec.corners, 8, corners, 8) == REL_OUT) continue;
    for (int ci = 0; ci < 8; ++ci) {
      NodeId cid = cur.get_child_id((uint8_t)ci);
      if (t.nodes.count(cid.v)) q.push_back(cid);
    }
    out->push_back(cur);
  }
}

// per-point culling
static bool frustum_contains(const double* clip_from_query, V3 p) {  // frustum.rs:120-125
  V3 c = transform_point(clip_from_query, p);
  double mn = std::fmin(std::fmin(c.x, c.y), c.z), mx = std::fmax(std::fmax(c.x, c.y), c.z);
  return mn > -1.0 && mx < 1.0;
}
static bool obb_contains(const Iso& obb_from_query, V3 h, V3 p) {  // obb.rs:83-90
  V3 q = iso_transform_point(obb_from_query, p);
  return std::fabs(q.x) <= h.x && std::fabs(q.y) <= h.y && std::fabs(q.z) <= h.z;
}
static bool aabb_contains(const double mn[3], const double mx[3], V3 p) {  // aabb.rs:46-48
  return mn[0] <= p.x && mn[1] <= p.y && mn[2] <= p.z && p.x < mx[0] && p.y < mx[1] && p.z < mx[2];
}

}  // namespace pcvq

using namespace pcvq;

extern "C" {

// ---- matrices ----
int pcvo_mat4_try_inverse(const double* m, double* out) { return try_inverse4(m, out) ? 1 : 0; }
void pcvo_perspective_new(double l, double r, double b, double t, double n, double f, double* m) { perspective_new(l, r, b, t, n, f, m); }
void pcvo_perspective_inverse(const double* p, double* m) { perspective_inverse(p, m); }
// nalgebra 0.22 Perspective3::new(aspect, fovy, znear, zfar).to_homogeneous() (geometry/perspective.rs)
void pcvo_perspective3_new(double aspect, double fovy, double znear, double zfar, double* m) {
  std::memset(m, 0, 16 * sizeof(double));
  at(m, 0, 0) = at(m, 1, 1) = at(m, 2, 2) = 1.0;
  at(m, 3, 2) = -1.0;
  double old_m22 = at(m, 1, 1);
  at(m, 1, 1) = 1.0 / std::tan(fovy / 2.0);           // set_fovy
  at(m, 0, 0) = at(m, 0, 0) * (at(m, 1, 1) / old_m22);
  at(m, 0, 0) = at(m, 1, 1) / aspect;                  // set_aspect
  at(m, 2, 2) = (zfar + znear) / (znear - zfar);       // set_znear_and_zfar
  at(m, 2, 3) = zfar * znear * 2.0 / (znear - zfar);
}
// Frustum::new (frustum.rs:101-108): iso = translation xyz + quaternion ijkw
void pcvo_frustum_new(const double iso7[7], const double* perspective, double* clip_from_query, double* query_from_clip) {
  Iso s{{iso7[0], iso7[1], iso7[2]}, {iso7[3], iso7[4], iso7[5], iso7[6]}};
  double inv_h[16], h[16], pinv[16];
  iso_to_homogeneous(iso_inverse(s), inv_h);
  iso_to_homogeneous(s, h);
  perspective_inverse(perspective, pinv);
  mat4_mul(perspective, inv_h, clip_from_query);
  mat4_mul(h, pinv, query_from_clip);
}
void pcvo_iso_transform_points(const double iso7[7], uint64_t n, const double* x, const double* y, const double* z,
                               double* ox, double* oy, double* oz) {
  Iso s{{iso7[0], iso7[1], iso7[2]}, {iso7[3], iso7[4], iso7[5], iso7[6]}};
  for (uint64_t i = 0; i < n; ++i) {
    V3 r = iso_transform_point(s, V3{x[i], y[i], z[i]});
    ox[i] = r.x;
    oy[i] = r.y;
    oz[i] = r.z;
  }
}

// ---- intersectors: kind 2 frustum (params = clip_from_query 16), 3 obb (t3, q4, half3), 1 aabb (min3, max3) ----
static bool make_intersector(int kind, const double* p, Intersector* s) {
  if (kind == 4) {  // Frustum::new: both matrices given (clip_from_query 16, query_from_clip 16)
    *s = frustum_intersector(p + 16);
  } else if (kind == 2) {
    double inv[16];
    if (!try_inverse4(p, inv)) return false;
    *s = frustum_intersector(inv);
  } else if (kind == 3) {
    Iso q{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}};
    *s = obb_intersector(q, V3{p[7], p[8], p[9]});
  } else {
    *s = aabb_intersector(p, p + 3);
  }
  return true;
}
// Cached axes for AABB targets: returns the number of axes (<= 26), -1 if the shape is invalid.
int pcvo_cached_axes(int kind, const double* params, double corners[24], double axes[78]) {
  Intersector s;
  if (!make_intersector(kind, params, &s)) return -1;
  Cached c = kind == 1 ? Cached{} : cache_for_aabb(s);
  if (kind == 1) {  // aabb.rs:98-107: the Aabb's own intersector uses just the three unit axes
    for (int i = 0; i < 8; ++i) c.corners[i] = s.corners[i];
    c.axes = {V3{1, 0, 0}, V3{0, 1, 0}, V3{0, 0, 1}};
  }
  for (int i = 0; i < 8; ++i) {
    corners[3 * i] = c.corners[i].x;
    corners[3 * i + 1] = c.corners[i].y;
    corners[3 * i + 2] = c.corners[i].z;
  }
  for (size_t i = 0; i < c.axes.size(); ++i) {
    axes[3 * i] = c.axes[i].x;
    axes[3 * i + 1] = c.axes[i].y;
    axes[3 * i + 2] = c.axes[i].z;
  }
  return (int)c.axes.size();
}
// Relation of each cube (min xyz, edge) against the shape's cached intersector; sizes optional (frustum matrix).
int pcvo_cull_cubes(int kind, const double* params, uint64_t m, const double* cubes4, uint8_t* relation, double* size_on_screen) {
  double corners[24], axes[78];
  int na = pcvo_cached_axes(kind, params, corners, axes);
  if (na < 0) return -1;
  for (uint64_t i = 0; i < m; ++i) {
    V3 cc[8];
    cube_corners(&cubes4[4 * i], cubes4[4 * i + 3], cc);
    relation[i] = (uint8_t)sat((const V3*)axes, na, (const V3*)corners, 8, cc, 8);
    if (size_on_screen) {
      bool ok;
      size_on_screen[i] = relative_size_on_screen(&cubes4[4 * i], cubes4[4 * i + 3], params, &ok);
      if (!ok) size_on_screen[i] = NAN;
    }
  }
  return 0;
}
// Generic Intersector::intersect of two shapes (sat.rs:146-152), used for the reference's own unit tests.
int pcvo_intersect_shapes(int kind_a, const double* pa, int kind_b, const double* pb) {
  Intersector a, b;
  if (!make_intersector(kind_a, pa, &a) || !make_intersector(kind_b, pb, &b)) return -1;
  return intersect(a, b);
}
// sat() on explicit data (axes n x 3, corners 8 x 3 each) — sat.rs tests
int pcvo_sat_raw(int naxes, const double* axes, const double* ca, int na, const double* cb, int nb) {
  return sat((const V3*)axes, naxes, (const V3*)ca, na, (const V3*)cb, nb);
}

// ---- octree queries: the octree is given as arrays (id halves, num_points); cubes derive from the bbox ----
static QOctree make_qoctree(const double bmin[3], const double bmax[3], uint64_t m, const uint64_t* hi, const uint64_t* lo,
                            const int64_t* num_points) {
  QOctree t;
  pcvo::Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  t.root_cube = pcvo::cube_bounding(b);
  for (uint64_t i = 0; i < m; ++i) {
    NodeId id = NodeId::from_high_low(hi[i], lo[i]);
    t.nodes[id.v] = QNode{num_points[i], id.find_bounding_cube(t.root_cube)};
  }
  return t;
}
// get_visible_nodes: writes ids in pop order; returns count, or -1 where the reference would panic.
int64_t pcvo_get_visible_nodes(const double bmin[3], const double bmax[3], uint64_t m, const uint64_t* hi, const uint64_t* lo,
                               const int64_t* num_points, const double* matrix, uint64_t* out_hi, uint64_t* out_lo) {
  QOctree t = make_qoctree(bmin, bmax, m, hi, lo, num_points);
  std::vector<NodeId> vis;
  if (!get_visible_nodes(t, matrix, &vis)) return -1;
  for (size_t i = 0; i < vis.size(); ++i) {
    out_hi[i] = vis[i].high();
    out_lo[i] = vis[i].low();
  }
  return (int64_t)vis.size();
}
int64_t pcvo_nodes_in_location(const double bmin[3], const double bmax[3], uint64_t m, const uint64_t* hi, const uint64_t* lo,
                               const int64_t* num_points, int kind, const double* params, uint64_t* out_hi, uint64_t* out_lo) {
  QOctree t = make_qoctree(bmin, bmax, m, hi, lo, num_points);
  std::vector<NodeId> out;
  if (kind == 0) {  // AllPoints: every node passes (math/mod.rs:139-160)
    Cached all;
    std::deque<NodeId> q{NodeId::root()};
    while (!q.empty()) {
      NodeId cur = q.front();
      q.pop_front();
      if (!t.nodes.count(cur.v)) continue;
      for (int ci = 0; ci < 8; ++ci)
        if (t.nodes.count(cur.get_child_id((uint8_t)ci).v)) q.push_back(cur.get_child_id((uint8_t)ci));
      out.push_back(cur);
    }
  } else {
    double corners[24], axes[78];
    int na = pcvo_cached_axes(kind, params, corners, axes);
    if (na < 0) return -1;
    Cached c;
    for (int i = 0; i < 8; ++i) c.corners[i] = {corners[3 * i], corners[3 * i + 1], corners[3 * i + 2]};
    for (int i = 0; i < na; ++i) c.axes.push_back({axes[3 * i], axes[3 * i + 1], axes[3 * i + 2]});
    nodes_in_location(t, c, &out);
  }
  for (size_t i = 0; i < out.size(); ++i) {
    out_hi[i] = out[i].high();
    out_lo[i] = out[i].low();
  }
  return (int64_t)out.size();
}

// ---- per-point keep mask (iterator.rs:96-119): shape AND optional closed interval on a f32 attribute ----
void pcvo_cull_points(int kind, const double* params, uint64_t n, const double* x, const double* y, const double* z,
                      const float* attr, const double* interval /* [lo, hi] or null */, uint8_t* keep) {
  Iso obb_from_query{};
  V3 h{};
  if (kind == 3) {
    Iso q{{params[0], params[1], params[2]}, {params[3], params[4], params[5], params[6]}};
    obb_from_query = iso_inverse(q);  // obb.rs:35-41
    h = {params[7], params[8], params[9]};
  }
  for (uint64_t i = 0; i < n; ++i) {
    V3 p{x[i], y[i], z[i]};
    bool k = true;
    if (kind == 1) k = aabb_contains(params, params + 3, p);
    else if (kind == 2 || kind == 4) k = frustum_contains(params, p);
    else if (kind == 3) k = obb_contains(obb_from_query, h, p);
    if (attr && interval) {
      double v = (double)attr[i];
      k = k && (interval[0] <= v && v <= interval[1]);  // math/mod.rs:86-88
    }
    keep[i] = k ? 1 : 0;
  }
}

// decode a node's .xyz bytes to f64 positions (raw.rs:141-215)
void pcvo_decode_positions(int enc, const double cube_min[3], double edge, uint64_t n, const uint8_t* xyz, double* x, double* y,
                           double* z) {
  int bpc = pcvo::bytes_per_coordinate((pcvo::Enc)enc);
  double* o[3] = {x, y, z};
  for (uint64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
      o[a][i] = pcvo::decode_coord((pcvo::Enc)enc, pcvo::get_le(xyz + (3 * i + (uint64_t)a) * (uint64_t)bpc, bpc), cube_min[a], edge);
}

}  // extern "C"
