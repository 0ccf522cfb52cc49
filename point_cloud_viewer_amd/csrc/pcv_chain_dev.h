// pcv_chain_dev.h — device-side exact arithmetic of the reference's per-level quantise -> decode chain.
//
// Everything here must round exactly like the reference's f64 code:
//   centre  = (min + (min + edge)) / 2                       src/geometry/aabb.rs:175-192
//   digit   = (p.x > c.x) << 2 | (p.y > c.y) << 1 | (p.z > c.z)   src/octree/node.rs:34-42
//   child   : edge /= 2 ; min += bit * edge                   src/octree/node.rs:157-172
//   encode  : t = clamp((p - min) / edge, 0, 1); u8/u16: trunc(max * t); f32: (float)t; f64: t
//                                                             src/read_write/codec.rs:102-121
//   decode  : (v / max).mul_add(edge, min)  |  v.mul_add(edge, min)   src/read_write/codec.rs:124-139
// The translation unit is compiled with -ffp-contract=off; the only fused operations are the explicit
// __fma_rn calls that restate `mul_add`. Divisions are IEEE-correct f64 divisions (no reciprocals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcv_internal.h"

// num::clamp semantics (NaN and -0.0 pass through) — needed verbatim for the float encodings.
__device__ __forceinline__ double pcv_clamp01(double t) { return (t < 0.0) ? 0.0 : ((t > 1.0) ? 1.0 : t); }

// Rust `as u8/u16` after the clamp: NaN -> 0, truncation toward zero; t <= 1 so no upper saturation.
__device__ __forceinline__ uint32_t pcv_fix_encode(double p, double mn, double edge, double maxval) {
  double t = (p - mn) / edge;
  // (t > 0 ? t : 0) maps NaN, -0.0 and negatives to 0 — same integer code as clamp + `as` cast.
  t = (t > 0.0) ? t : 0.0;
  t = (t > 1.0) ? 1.0 : t;
  return (uint32_t)(maxval * t);
}

// Raw code (integer value or IEEE bit pattern) of one coordinate.
__device__ __forceinline__ uint64_t pcv_encode_coord(uint32_t enc, double p, double mn, double edge) {
  switch (enc) {
    case PCV_ENC_UINT8: return pcv_fix_encode(p, mn, edge, 255.0);
    case PCV_ENC_UINT16: return pcv_fix_encode(p, mn, edge, 65535.0);
    case PCV_ENC_FLOAT32: {
      float f = (float)pcv_clamp01((p - mn) / edge);  // round-to-nearest-even
      return (uint64_t)__float_as_uint(f);
    }
    default: return (uint64_t)__double_as_longlong(pcv_clamp01((p - mn) / edge));
  }
}

__device__ __forceinline__ double pcv_decode_coord(uint32_t enc, uint64_t code, double mn, double edge) {
  switch (enc) {
    case PCV_ENC_UINT8: return __fma_rn((double)(uint32_t)code / 255.0, edge, mn);
    case PCV_ENC_UINT16: return __fma_rn((double)(uint32_t)code / 65535.0, edge, mn);
    case PCV_ENC_FLOAT32: return __fma_rn((double)__uint_as_float((uint32_t)code), edge, mn);
    default: return __fma_rn(__longlong_as_double((long long)code), edge, mn);
  }
}

// One level of the chain for one coordinate: returns the octant bit, moves `mn` to the child cube,
// replaces `p` by its encode->decode image in the child cube and reports the code.
__device__ __forceinline__ uint32_t pcv_chain_coord(uint32_t enc, double e_parent, double e_child, double& p,
                                                    double& mn, uint64_t& code) {
  double mx = mn + e_parent;
  double c = (mn + mx) / 2.0;
  uint32_t bit = p > c ? 1u : 0u;
  mn = mn + (bit ? e_child : 0.0);  // `bit as f64 * edge` is exactly e or +0.0
  code = pcv_encode_coord(enc, p, mn, e_child);
  p = pcv_decode_coord(enc, code, mn, e_child);
  return bit;
}
