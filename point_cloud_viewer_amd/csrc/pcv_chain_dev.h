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
// __fma_rn calls that restate `mul_add` and the FMA residuals of the exact constant-divisor division below, whose
// results are bit-identical to IEEE f64 division (no approximate reciprocals anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcv_internal.h"

// Correctly rounded x / e for a divisor that is constant across the grid, without the ~12-instruction IEEE
// division expansion. The host supplies the reciprocal as a double-double: yh = RN(1/e), yl = RN(1/e - yh).
//   q0 = RN(x*yh + RN(x*yl))   one rounding of x * (yh + yl) up to ~2^-105 relative: a faithful quotient (< 1 ulp)
//   r  = x - e*q0              exact (one FMA), because q0 is faithful
//   q  = RN(q0 + r*yh)         == RN(x/e) by Markstein's theorem (Markstein 1990; Muller et al., Handbook of
//                              Floating-Point Arithmetic, "division with an FMA"): q0 faithful, yh correctly rounded
// Four f64 operations. The theorem needs no overflow/underflow in the residual: outside 2^-900 <= |x| <= 2^900 (also
// x == 0 for the sign of zero, inf, NaN) and for divisors outside [2^-100, 2^100] (yh == 0 from the host) the IEEE
// division is used instead. `pcv_selftest_division` checks the routine bit-for-bit against IEEE division on the device.
// GUARD = false drops the range check; only valid where the caller has established that x is finite with
// |x| <= 2^900 and yh != 0, and only for the integer encodings (where a zero or tiny x gives code 0 either way).
struct PcvRecip {
  double hi, lo;
};
template <bool GUARD = true>
__device__ __forceinline__ double pcv_div_const(double x, double e, PcvRecip y) {
  const double q0 = __fma_rn(x, y.hi, x * y.lo);
  const double r = __fma_rn(-e, q0, x);
  double q = __fma_rn(r, y.hi, q0);
  if (!GUARD) return q;
  const double ax = fabs(x);
  if (__builtin_expect(!(ax >= 0x1p-900 && ax <= 0x1p+900) || y.hi == 0.0, 0)) {
    // The empty volatile asm pins the IEEE division inside this (almost never taken) branch: without it the
    // compiler if-converts the branch and executes the full division expansion for every lane.
    double xs = x;
    asm volatile("" : "+v"(xs));
    q = xs / e;
  }
  return q;
}
// v / maxval for an integer code v in [0, maxval], maxval = 255 or 65535: RN(v*yh + RN(v*yl)) with the double-double
// reciprocal of maxval is bit-identical to IEEE division for every code (exhaustive: tests/test_oracle_kats.py on the
// host with exact rational FMA emulation, pcv_selftest_division on the device). Two f64 operations.
#define PCV_RECIP_255 PcvRecip{0x1.0101010101010p-8, 0x1.0101010101010p-64}
#define PCV_RECIP_65535 PcvRecip{0x1.0001000100010p-16, 0x1.0001000100010p-80}
__device__ __forceinline__ double pcv_div_code(double v, PcvRecip y) { return __fma_rn(v, y.hi, v * y.lo); }

// num::clamp semantics (NaN and -0.0 pass through) — needed verbatim for the float encodings.
__device__ __forceinline__ double pcv_clamp01(double t) { return (t < 0.0) ? 0.0 : ((t > 1.0) ? 1.0 : t); }

// t = num::clamp((p - mn) / edge, 0, 1) for the float encodings, where the exact t matters down to the sign of zero.
// Wherever x = p - mn is nonzero, finite and moderate (2^-900 <= |x| <= 2^900, tame divisor) the exact constant-divisor
// division applies and its quotient is finite and NONZERO — and for such t num::clamp is min(max(t, 0), 1): two
// instructions (the compiler folds them into the clamp modifier of the division's last FMA) instead of two compares
// and four selects per coordinate. Zeros (sign!), denormals, infinities, NaN and untamed divisors take the IEEE
// division and the literal clamp, out of line.
// TAME: the caller has established once per point that the coordinates are finite with |v| <= 2^500 and that the level
// table is tame (PcvLevels::fast_ok: every inv_edge.hi != 0, |cube min| <= 2^500 + edges): |x| <= 2^900 and the divisor
// check then hold by construction and ONE comparison — "not (|x| >= 2^-900)", which also catches the zeros whose sign the
// Markstein step would lose — is all that guards the fast path (round 4: a Float32-coded level spent 9 of its 48 f64-pipe
// instructions on the three-part range test).
template <bool TAME>
__device__ __forceinline__ double pcv_unit_quotient_t(double p, double mn, double edge, PcvRecip inv_edge) {
  const double x = p - mn;
  const double ax = fabs(x);
  const bool slow = TAME ? !(ax >= 0x1p-900) : (!(ax >= 0x1p-900 && ax <= 0x1p+900) || inv_edge.hi == 0.0);
  if (__builtin_expect(slow, 0)) {
    double xs = x;
    asm volatile("" : "+v"(xs));  // pins the division expansion inside this (almost never taken) branch
    return pcv_clamp01(xs / edge);
  }
  return fmin(fmax(pcv_div_const<false>(x, edge, inv_edge), 0.0), 1.0);
}
__device__ __forceinline__ double pcv_unit_quotient(double p, double mn, double edge, PcvRecip inv_edge) {
  return pcv_unit_quotient_t<false>(p, mn, edge, inv_edge);
}

// Rust `as u8/u16` after the clamp: NaN -> 0, truncation toward zero; t <= 1 so no upper saturation.
template <bool GUARD = true>
__device__ __forceinline__ uint32_t pcv_fix_encode(double p, double mn, double edge, PcvRecip inv_edge, double maxval) {
  double t = pcv_div_const<GUARD>(p - mn, edge, inv_edge);
  // maxNum(t, 0) maps NaN (t is never a signalling NaN: it is an arithmetic result), -0.0 and negatives to a zero —
  // same integer code as num::clamp + `as` cast; minNum(t, 1) is the upper clamp. Two instructions instead of six.
  t = fmax(t, 0.0);
  t = fmin(t, 1.0);
  return (uint32_t)(maxval * t);
}

// Raw code (integer value or IEEE bit pattern) of one coordinate.
template <bool GUARD = true>
__device__ __forceinline__ uint64_t pcv_encode_coord(uint32_t enc, double p, double mn, double edge, PcvRecip inv_edge) {
  switch (enc) {
    case PCV_ENC_UINT8: return pcv_fix_encode<GUARD>(p, mn, edge, inv_edge, 255.0);
    case PCV_ENC_UINT16: return pcv_fix_encode<GUARD>(p, mn, edge, inv_edge, 65535.0);
    case PCV_ENC_FLOAT32: {
      float f = (float)pcv_unit_quotient(p, mn, edge, inv_edge);  // round-to-nearest-even
      return (uint64_t)__float_as_uint(f);
    }
    default: return (uint64_t)__double_as_longlong(pcv_unit_quotient(p, mn, edge, inv_edge));
  }
}

__device__ __forceinline__ double pcv_decode_coord(uint32_t enc, uint64_t code, double mn, double edge) {
  switch (enc) {
    case PCV_ENC_UINT8: return __fma_rn(pcv_div_code((double)(uint32_t)code, PCV_RECIP_255), edge, mn);
    case PCV_ENC_UINT16: return __fma_rn(pcv_div_code((double)(uint32_t)code, PCV_RECIP_65535), edge, mn);
    case PCV_ENC_FLOAT32: return __fma_rn((double)__uint_as_float((uint32_t)code), edge, mn);
    default: return __fma_rn(__longlong_as_double((long long)code), edge, mn);
  }
}

// Inside the chain the code is kept in the value domain (a double): the integer value for u8/u16, (double)(float)t
// for Float32, t for Float64 — no int conversions per level; pcv_val_to_code makes the raw bits once at the end.
template <int ENC, bool GUARD>
__device__ __forceinline__ double pcv_encode_val(double p, double mn, double edge, PcvRecip inv_edge) {
  if (ENC == PCV_ENC_UINT8 || ENC == PCV_ENC_UINT16) {
    const double maxval = ENC == PCV_ENC_UINT8 ? 255.0 : 65535.0;
    double t = pcv_div_const<GUARD>(p - mn, edge, inv_edge);
    t = fmax(t, 0.0);  // NaN, -0.0, negatives -> zero (see pcv_fix_encode)
    t = fmin(t, 1.0);
    return trunc(maxval * t);  // Rust `as u8/u16`: truncation toward zero, exact in f64
  }
  const double t = pcv_unit_quotient_t<!GUARD>(p, mn, edge, inv_edge);  // GUARD = false: tame point, tame table
  return ENC == PCV_ENC_FLOAT32 ? (double)(float)t : t;
}
template <int ENC>
__device__ __forceinline__ double pcv_decode_val(double cd, double mn, double edge) {
  if (ENC == PCV_ENC_UINT8) return __fma_rn(pcv_div_code(cd, PCV_RECIP_255), edge, mn);
  if (ENC == PCV_ENC_UINT16) return __fma_rn(pcv_div_code(cd, PCV_RECIP_65535), edge, mn);
  return __fma_rn(cd, edge, mn);
}
__device__ __forceinline__ uint64_t pcv_val_to_code(uint32_t enc, double cd) {
  switch (enc) {
    case PCV_ENC_UINT8:
    case PCV_ENC_UINT16: return (uint64_t)(uint32_t)cd;
    case PCV_ENC_FLOAT32: return (uint64_t)__float_as_uint((float)cd);
    default: return (uint64_t)__double_as_longlong(cd);
  }
}

// `min += bit as f64 * edge` (node.rs:161-170): the product is exactly `edge` or +0.0, so a single FMA with the bit as
// 1.0 / 0.0 rounds identically (one select for the high word of the factor + the FMA instead of two selects + an add).
__device__ __forceinline__ double pcv_step_min(double mn, bool bit, double ec) {
  return __fma_rn(__hiloint2double(bit ? 0x3ff00000 : 0, 0), ec, mn);
}

// One level of the chain for one coordinate with the encoding known at compile time: returns the octant bit,
// moves `mn` to the child cube, replaces `p` by its encode->decode image in the child cube and reports the code
// (value domain).
template <int ENC, bool GUARD>
__device__ __forceinline__ uint32_t pcv_chain_coord_t(double e_parent, double e_child, PcvRecip inv_e_child, double& p,
                                                      double& mn, double& cd) {
  const double mx = mn + e_parent;
  const double c = (mn + mx) / 2.0;
  const uint32_t bit = p > c ? 1u : 0u;
  mn = pcv_step_min(mn, bit != 0u, e_child);
  cd = pcv_encode_val<ENC, GUARD>(p, mn, e_child, inv_e_child);
  p = pcv_decode_val<ENC>(cd, mn, e_child);
  return bit;
}

// One level for all three coordinates as a single straight-line block (the encoding switch is taken once per
// level and is wave-uniform, so the three dependency chains interleave). Returns the octant digit.
template <int ENC, bool GUARD>
__device__ __forceinline__ uint32_t pcv_chain_level_t(double ep, double ec, PcvRecip ic, double& px, double& py, double& pz,
                                                      double& mx, double& my, double& mz, double& cx, double& cy,
                                                      double& cz) {
  const uint32_t bx = pcv_chain_coord_t<ENC, GUARD>(ep, ec, ic, px, mx, cx);
  const uint32_t by = pcv_chain_coord_t<ENC, GUARD>(ep, ec, ic, py, my, cy);
  const uint32_t bz = pcv_chain_coord_t<ENC, GUARD>(ep, ec, ic, pz, mz, cz);
  return (bx << 2) | (by << 1) | bz;
}
// GUARD = false: the caller checked once per point that the coordinates are finite and moderate (pcv_point_is_tame)
// and the level table is tame (PcvLevels::fast_ok); the integer-encoded levels then run without per-division checks.
template <bool GUARD>
__device__ __forceinline__ uint32_t pcv_chain_level(uint32_t enc, double ep, double ec, PcvRecip ic, double& px, double& py,
                                                    double& pz, double& mx, double& my, double& mz, double& cx,
                                                    double& cy, double& cz) {
  switch (enc) {
    case PCV_ENC_UINT8: return pcv_chain_level_t<PCV_ENC_UINT8, GUARD>(ep, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    case PCV_ENC_UINT16: return pcv_chain_level_t<PCV_ENC_UINT16, GUARD>(ep, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    case PCV_ENC_FLOAT32: return pcv_chain_level_t<PCV_ENC_FLOAT32, GUARD>(ep, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    default: return pcv_chain_level_t<PCV_ENC_FLOAT64, GUARD>(ep, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
  }
}

// The same level in two halves, for callers that want the digit BEFORE the (long) encode/decode arithmetic — the
// single-chain pass issues the load of the child's walk record right after the digit, so that its latency hides behind
// the ~55 f64 operations of the rest of the level. Same operations, same order per coordinate, same results.
__device__ __forceinline__ uint32_t pcv_chain_digit(double e_parent, double px, double py, double pz, double mx, double my,
                                                    double mz) {
  const double cx = (mx + (mx + e_parent)) / 2.0, cy = (my + (my + e_parent)) / 2.0, cz = (mz + (mz + e_parent)) / 2.0;
  return (px > cx ? 4u : 0u) | (py > cy ? 2u : 0u) | (pz > cz ? 1u : 0u);
}
// The same digit from the integer codes of the parent level, without touching the cube: for a u8 / u16-coded level k
// with codes c in [0, M] (M = 255 / 65535, odd) the position the next comparison sees is p = fma(RN(c / M), e, mn) and
// the centre is c0 = fl(fl(mn + fl(mn + e)) / 2). Exactly, P* = mn + (c / M) e and C* = mn + e / 2 differ by
// e |c / M - 1 / 2| >= e / (2 M) — there is no tie, M is odd. Rounding moves p by at most u (|mn| + 2 e) and c0 by at
// most u (1.5 |mn| + e) (u = 2^-53, standard model, no under/overflow for tame tables). So wherever
//     1.01 u (2.5 A / e + 3) < 1 / (2 M),     A = the largest |coordinate| of the root cube,
// (p > c0) == (c > M / 2) == (c > 127 resp. 32767) for every point: one compare per coordinate instead of two
// additions, a multiplication and a compare. pcv_make_levels checks the inequality per level with a factor of two in
// hand (PcvLevels::digit_half; ECEF coordinates with millimetre cubes pass) and tests/test_oracle_kats.py replays the
// boundary codes at the admitted |mn| / e ratios in exact rational arithmetic. Only used on the unguarded path (finite,
// moderate inputs); Float32 / Float64-coded parents (ties at t = 0.5) and the root keep the comparison above.
__device__ __forceinline__ uint32_t pcv_digit_from_codes(double half, double vx, double vy, double vz) {
  return (vx > half ? 4u : 0u) | (vy > half ? 2u : 0u) | (vz > half ? 1u : 0u);
}
template <int ENC, bool GUARD>
__device__ __forceinline__ void pcv_chain_apply_t(uint32_t d, double ec, PcvRecip ic, double& px, double& py, double& pz, double& mx,
                                                  double& my, double& mz, double& cx, double& cy, double& cz) {
  mx = pcv_step_min(mx, (d & 4u) != 0u, ec);
  my = pcv_step_min(my, (d & 2u) != 0u, ec);
  mz = pcv_step_min(mz, (d & 1u) != 0u, ec);
  cx = pcv_encode_val<ENC, GUARD>(px, mx, ec, ic);
  cy = pcv_encode_val<ENC, GUARD>(py, my, ec, ic);
  cz = pcv_encode_val<ENC, GUARD>(pz, mz, ec, ic);
  px = pcv_decode_val<ENC>(cx, mx, ec);
  py = pcv_decode_val<ENC>(cy, my, ec);
  pz = pcv_decode_val<ENC>(cz, mz, ec);
}
template <bool GUARD>
__device__ __forceinline__ void pcv_chain_apply(uint32_t enc, uint32_t d, double ec, PcvRecip ic, double& px, double& py, double& pz,
                                                double& mx, double& my, double& mz, double& cx, double& cy, double& cz) {
  switch (enc) {
    case PCV_ENC_UINT8: return pcv_chain_apply_t<PCV_ENC_UINT8, GUARD>(d, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    case PCV_ENC_UINT16: return pcv_chain_apply_t<PCV_ENC_UINT16, GUARD>(d, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    case PCV_ENC_FLOAT32: return pcv_chain_apply_t<PCV_ENC_FLOAT32, GUARD>(d, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    default: return pcv_chain_apply_t<PCV_ENC_FLOAT64, GUARD>(d, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
  }
}

// ---- the same level with the octant bits kept as three booleans (round 4) ------------------------------------------
// The digit only exists to index the child: the three comparisons produce lane masks, `min += bit * edge` selects the
// high word of its 1.0 / 0.0 factor straight from the mask (one v_cndmask per coordinate) and the child index is assembled
// from the same masks — instead of packing the digit and taking it apart again (six more integer instructions per level).
struct PcvOctBits {
  bool x, y, z;
  __device__ __forceinline__ uint32_t digit() const { return (x ? 4u : 0u) | (y ? 2u : 0u) | (z ? 1u : 0u); }
};
__device__ __forceinline__ PcvOctBits pcv_chain_bits(double e_parent, double px, double py, double pz, double mx, double my, double mz) {
  const double cx = (mx + (mx + e_parent)) / 2.0, cy = (my + (my + e_parent)) / 2.0, cz = (mz + (mz + e_parent)) / 2.0;
  return PcvOctBits{px > cx, py > cy, pz > cz};
}
// Integer codes of a u8 / u16-coded parent level: see pcv_digit_from_codes.
__device__ __forceinline__ PcvOctBits pcv_bits_from_codes(double half, double vx, double vy, double vz) {
  return PcvOctBits{vx > half, vy > half, vz > half};
}
// Float32 codes of the parent level (round 4). The position the comparison sees is p = fma(v, e, mn) with v = (double)(float)t,
// the centre c0 = fl(fl(mn + fl(mn + e)) / 2). Exactly, P* - C* = (v - 1/2) e, and the floats next to 1/2 are 1/2 - 2^-25 and
// 1/2 + 2^-24: unless v == 1/2, |P* - C*| >= 2^-25 e. Rounding moves p by at most u (|mn| + e) and c0 by at most
// u (1.5 |mn| + e) (u = 2^-53), so wherever
//     1.01 u (2.5 A / e + 3) < 2^-25,     A = the largest |coordinate| of the root cube,
// (p > c0) == (v > 1/2) for every v != 1/2 — the same inequality as pcv_digit_from_codes with M = 2^24, checked per level by
// pcv_make_levels with a factor of two in hand (PcvLevels::digit_mode == 2). A code of exactly 1/2 is a genuine tie of the
// exact values: there the rounded comparison decides, so a wave that holds such a lane takes the comparison against the
// centre for that level (wave-uniform branch; one point in 2^24 per coordinate for smooth data).
__device__ __forceinline__ bool pcv_f32_code_tie(double vx, double vy, double vz) { return vx == 0.5 || vy == 0.5 || vz == 0.5; }

template <int ENC, bool GUARD>
__device__ __forceinline__ void pcv_chain_apply_bits_t(PcvOctBits b, double ec, PcvRecip ic, double& px, double& py, double& pz, double& mx,
                                                       double& my, double& mz, double& cx, double& cy, double& cz) {
  mx = pcv_step_min(mx, b.x, ec);
  my = pcv_step_min(my, b.y, ec);
  mz = pcv_step_min(mz, b.z, ec);
  if constexpr (!GUARD && (ENC == PCV_ENC_FLOAT32 || ENC == PCV_ENC_FLOAT64)) {
    // float encodings of a tame point in a tame table (pcv_unit_quotient_t<true>), the three coordinates behind ONE branch:
    // a lane whose dividends are all >= 2^-900 in magnitude takes the constant-divisor division for all three; any other
    // lane (a zero whose sign matters, a denormal) takes the IEEE division for all three — it is correct everywhere, and
    // one branch scaffold per level replaces three
    const double x = px - mx, y = py - my, z = pz - mz;
    double tx, ty, tz;
    if (__builtin_expect(!(fabs(x) >= 0x1p-900 && fabs(y) >= 0x1p-900 && fabs(z) >= 0x1p-900), 0)) {
      double xs = x, ys = y, zs = z;
      asm volatile("" : "+v"(xs), "+v"(ys), "+v"(zs));  // pins the division expansions inside this (almost never taken) branch
      tx = pcv_clamp01(xs / ec), ty = pcv_clamp01(ys / ec), tz = pcv_clamp01(zs / ec);
    } else {
      tx = fmin(fmax(pcv_div_const<false>(x, ec, ic), 0.0), 1.0);
      ty = fmin(fmax(pcv_div_const<false>(y, ec, ic), 0.0), 1.0);
      tz = fmin(fmax(pcv_div_const<false>(z, ec, ic), 0.0), 1.0);
    }
    cx = ENC == PCV_ENC_FLOAT32 ? (double)(float)tx : tx;
    cy = ENC == PCV_ENC_FLOAT32 ? (double)(float)ty : ty;
    cz = ENC == PCV_ENC_FLOAT32 ? (double)(float)tz : tz;
    px = pcv_decode_val<ENC>(cx, mx, ec);
    py = pcv_decode_val<ENC>(cy, my, ec);
    pz = pcv_decode_val<ENC>(cz, mz, ec);
    return;
  }
  cx = pcv_encode_val<ENC, GUARD>(px, mx, ec, ic);
  cy = pcv_encode_val<ENC, GUARD>(py, my, ec, ic);
  cz = pcv_encode_val<ENC, GUARD>(pz, mz, ec, ic);
  px = pcv_decode_val<ENC>(cx, mx, ec);
  py = pcv_decode_val<ENC>(cy, my, ec);
  pz = pcv_decode_val<ENC>(cz, mz, ec);
}
// The integer-coded levels of a tame point in a tame table with the level's maximum code (255 / 65535) and its double-double
// reciprocal as SCALARS: the same operations as pcv_chain_apply_bits_t<PCV_ENC_UINT8 / UINT16, false>, so that the u16- and
// the u8-coded levels of the chain pass share one loop body (round 5).
__device__ __forceinline__ void pcv_chain_apply_bits_int(PcvOctBits b, double ec, PcvRecip ic, double maxval, PcvRecip rmax, double& px,
                                                         double& py, double& pz, double& mx, double& my, double& mz, double& cx,
                                                         double& cy, double& cz) {
  mx = pcv_step_min(mx, b.x, ec);
  my = pcv_step_min(my, b.y, ec);
  mz = pcv_step_min(mz, b.z, ec);
  cx = trunc(maxval * fmin(fmax(pcv_div_const<false>(px - mx, ec, ic), 0.0), 1.0));  // pcv_encode_val<integer, false>
  cy = trunc(maxval * fmin(fmax(pcv_div_const<false>(py - my, ec, ic), 0.0), 1.0));
  cz = trunc(maxval * fmin(fmax(pcv_div_const<false>(pz - mz, ec, ic), 0.0), 1.0));
  px = __fma_rn(pcv_div_code(cx, rmax), ec, mx);  // pcv_decode_val<integer>
  py = __fma_rn(pcv_div_code(cy, rmax), ec, my);
  pz = __fma_rn(pcv_div_code(cz, rmax), ec, mz);
}

template <bool GUARD>
__device__ __forceinline__ void pcv_chain_apply_bits(uint32_t enc, PcvOctBits b, double ec, PcvRecip ic, double& px, double& py, double& pz,
                                                     double& mx, double& my, double& mz, double& cx, double& cy, double& cz) {
  switch (enc) {
    case PCV_ENC_UINT8: return pcv_chain_apply_bits_t<PCV_ENC_UINT8, GUARD>(b, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    case PCV_ENC_UINT16: return pcv_chain_apply_bits_t<PCV_ENC_UINT16, GUARD>(b, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    case PCV_ENC_FLOAT32: return pcv_chain_apply_bits_t<PCV_ENC_FLOAT32, GUARD>(b, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
    default: return pcv_chain_apply_bits_t<PCV_ENC_FLOAT64, GUARD>(b, ec, ic, px, py, pz, mx, my, mz, cx, cy, cz);
  }
}

// Start of a point's chain. Raw points start at the root (returns level 1). Routed points (PcvRouted) arrive as their
// level-1 state: the octant digit selects the level-1 cube with the same `min += bit * edge` step the chain uses, the
// Float32 codes decode to exactly the position the sending rank held after level 1 (returns level 2, digit in d1,
// codes in the value domain in vx..vz).
__device__ __forceinline__ int pcv_chain_start(const PcvLevels& lv, const PcvRouted& r, const double* __restrict__ x,
                                               const double* __restrict__ y, const double* __restrict__ z, uint64_t src,
                                               double& px, double& py, double& pz, double& mx, double& my, double& mz,
                                               double& vx, double& vy, double& vz, uint32_t& d1) {
  mx = lv.root_min[0], my = lv.root_min[1], mz = lv.root_min[2];
  if (!r.oct) {
    px = x[src], py = y[src], pz = z[src];
    d1 = 0;
    return 1;
  }
  d1 = r.oct[src * r.oct_stride] & 7u;
  const double e1 = lv.edge[1];
  mx = pcv_step_min(mx, (d1 & 4u) != 0u, e1);
  my = pcv_step_min(my, (d1 & 2u) != 0u, e1);
  mz = pcv_step_min(mz, (d1 & 1u) != 0u, e1);
  vx = (double)__uint_as_float(r.cx[src]);
  vy = (double)__uint_as_float(r.cy[src]);
  vz = (double)__uint_as_float(r.cz[src]);
  px = pcv_decode_val<PCV_ENC_FLOAT32>(vx, mx, e1);
  py = pcv_decode_val<PCV_ENC_FLOAT32>(vy, my, e1);
  pz = pcv_decode_val<PCV_ENC_FLOAT32>(vz, mz, e1);
  return 2;
}

// One check per point instead of one per division: finite and |v| <= 2^500 keeps every p_k - m_k of the chain
// finite and below 2^900 (cube mins and edges are bounded by PcvLevels::fast_ok on the host).
__device__ __forceinline__ bool pcv_point_is_tame(double x, double y, double z) {
  return fabs(x) <= 0x1p+500 && fabs(y) <= 0x1p+500 && fabs(z) <= 0x1p+500;  // false for NaN / inf
}
