// pcv_encode.hip — K5 leaf_encode (input order) and K6 promote_encode (sorted order).
//
// K5: for every input point replay the quantise->decode chain, walking the node table with the digits it
//     produces until a leaf is reached (generation.rs:78-99,167-177) and emit the record
//     (leaf rank, leaf-level codes, colour, intensity) that the stable record sort groups by leaf.
// K6: closed form of the bottom-up `i % 8 == 0` promotion (generation.rs:195-253,335-387):
//     a point at position j of its node's stream climbs while j % 8 == 0 (new j = offset of the child
//     inside the parent's stream + j / 8), each climb re-encoding decode_k -> encode_{k-1}
//     (generation.rs:222-238); a point that stays in a non-root node is rewritten once
//     (encode_k(decode_k(b))) at slot j - j/8 - 1; the root keeps everything it receives.
//     Output is written node-contiguous: exactly the bytes of <node>.xyz/.rgb/.intensity.
#include <algorithm>
#include <cstdlib>

#include "pcv_chain_dev.h"
#include "pcv_settle_dev.h"
#include "pcv_spec.h"

#ifndef PCV_KEEP_BRANCH
#define PCV_KEEP_BRANCH 0
#endif
#ifndef PCV_ENC_SEGMENTS
#define PCV_ENC_SEGMENTS 1
#endif

namespace {

__global__ __launch_bounds__(256) void leaf_encode_kernel(
    PcvLevels lv, const uint64_t* __restrict__ walk, uint64_t n, const double* __restrict__ x,
    const double* __restrict__ y, const double* __restrict__ z, PcvRouted routed, const uint8_t* __restrict__ color,
    uint32_t color_stride, const float* __restrict__ intensity, uint32_t* __restrict__ rank, uint4* __restrict__ payload,
    uint32_t* __restrict__ cx_hi, uint32_t* __restrict__ cy_hi, uint32_t* __restrict__ cz_hi,
    uint32_t* __restrict__ inten_bits) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // Walk the node table while replaying the chain: the digit of level L comes out of the chain step itself
  // (identical to K2's digit by construction), so no key has to be read back.
  // walk record = first_child | child_mask << 32 | leaf << 40 | level << 48 (leaves: low 32 bits = leaf rank)
  uint64_t rec = walk[0];
  double px, py, pz, mx, my, mz;
  double vx = 0, vy = 0, vz = 0;
  uint32_t d1;
  int L = 0;
  bool done = false;
  if (pcv_chain_start(lv, routed, x, y, z, i, px, py, pz, mx, my, mz, vx, vy, vz, d1) == 2) {
    // routed input: level 1 is given (digit + codes), take the step through the node table only
    L = 1;
    const uint32_t mask = (uint32_t)(rec >> 32) & 0xffu;
    rec = walk[(uint32_t)rec + __popc(mask & ((1u << d1) - 1u))];
    done = ((rec >> 40) & 1ull) || L >= lv.nlevels;
  }
  if (!done) {
    if (lv.fast_ok && pcv_point_is_tame(px, py, pz)) {
      do {
        ++L;
        const uint32_t d = pcv_chain_level<false>(lv.enc[L], lv.edge[L - 1], lv.edge[L], PcvRecip{lv.inv_edge[L], lv.inv_edge_lo[L]}, px, py, pz, mx, my, mz, vx, vy, vz);
        const uint32_t mask = (uint32_t)(rec >> 32) & 0xffu;
        rec = walk[(uint32_t)rec + __popc(mask & ((1u << d) - 1u))];
      } while (!((rec >> 40) & 1ull) && L < lv.nlevels);
    } else {
      do {
        ++L;
        const uint32_t d = pcv_chain_level<true>(lv.enc[L], lv.edge[L - 1], lv.edge[L], PcvRecip{lv.inv_edge[L], lv.inv_edge_lo[L]}, px, py, pz, mx, my, mz, vx, vy, vz);
        const uint32_t mask = (uint32_t)(rec >> 32) & 0xffu;
        rec = walk[(uint32_t)rec + __popc(mask & ((1u << d) - 1u))];
      } while (!((rec >> 40) & 1ull) && L < lv.nlevels);
    }
  }
  rank[i] = (uint32_t)rec;
  const uint32_t leaf_enc = lv.enc[L];
  const uint64_t ccx = pcv_val_to_code(leaf_enc, vx), ccy = pcv_val_to_code(leaf_enc, vy), ccz = pcv_val_to_code(leaf_enc, vz);
  const uint8_t* c = color + i * color_stride;
  payload[i] = make_uint4((uint32_t)ccx, (uint32_t)ccy, (uint32_t)ccz,
                          (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16));
  if (cx_hi) {  // some leaf level is Float64-encoded: carry the high words too
    cx_hi[i] = (uint32_t)(ccx >> 32);
    cy_hi[i] = (uint32_t)(ccy >> 32);
    cz_hi[i] = (uint32_t)(ccz >> 32);
  }
  if (inten_bits) inten_bits[i] = __float_as_uint(intensity[i]);
}

// ---- the record a point leaves the chain pass with (both forms of the pass) -----------------------------------------------
// 12-byte record: key = predicted leaf << 8 | blue, payload = {cx | cy << 16, cz | red << 16 | green << 24}. The codes of a
// Float32-coded level do not fit 16 bits: they go to an entry of the dense `wide` POOL and the record names that entry.
// Pool entries are handed out per WAVE (round 4): the lanes whose record level is Float32-coded are counted with one
// ballot, the wave's first such lane reserves that many consecutive entries with ONE returning atomic on the counter of the
// pool region its input slice belongs to (pcv_internal.h: kPcvPoolRegions regions — one counter for everything serialises
// the whole pass), and lane k of them takes entry base + k. Waves without such a lane (most of them: the deal below groups points by depth,
// and only the shallow levels are Float32-coded) pay a ballot and a scalar branch. A leaf's records meet `settle` in input
// order, i.e. in runs of entries that one wave wrote side by side: the 4.7 M entries of the bench cloud are 75 MB
// (Infinity-Cache resident) read in runs, where round 3 gathered one 16-byte entry per HBM line out of a 1.6 GB array indexed
// by input position (settle: 1.55 x its algorithmic traffic). Which entry a point gets depends on the order the waves reach
// the atomic — it does not matter: the record names its entry, nothing else refers to it.
__device__ __forceinline__ uint32_t pcv_load_rgb(const uint8_t* __restrict__ c, bool room) {
  // r | g << 8 | b << 16: one unaligned 32-bit load where a fourth byte exists behind the colour (every point but the last)
  if (room) {
    uint32_t w;
    __builtin_memcpy(&w, c, 4);
    return w & 0xffffffu;
  }
  return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
}
// ---- single-chain build (pcv_spec.h): ONE chain pass down the predicted tree T'' --------------------------------------
// Every inner node of T'' has all eight children (consecutive walk records), so a digit indexes the child directly.
// A point passing through a candidate node (sampled count close to the capacity) leaves this pass with the codes it has
// AT that node (the first such node on its path) instead of those of its predicted leaf: a point's chain from level k on
// is a function of its level-k codes alone, so if the exact counts later say the candidate is a leaf those ARE the leaf
// codes, and if it is split the chain is continued from them (spec_continue_kernel) once the record sort has made the
// true leaves contiguous. Output: predicted-leaf rank (the exact counts need it either way) + payload {codes, rgb}.
//
// Depth binning (BIN): lanes of one wave replay the chain until the DEEPEST of their 64 points reaches its leaf. In
// input order that is 9.7 levels per wave for a mean leaf depth of 6.7 (config-2 cloud) — 45 % of the f64 work runs
// with the lane already finished. So the workgroup first PREDICTS every point's leaf depth — one byte lookup in a
// 128^3 grid over the root cube that spec_depth_grid_kernel fills from T'' (depth of the leaf that covers the cell, 8 =
// "deeper than the grid") — and re-deals its points so that every wave gets points of (nearly) the same depth, deepest
// waves first. The prediction is a hint: a wrong guess costs time, never changes a result — the exact chain below
// still does all the work. Coordinates travel through LDS; outputs go to the point's own index.
constexpr int kSpecClasses = 24;  // predicted depth 0..21 (+ padding lanes)
#ifndef PCV_GRID_BITS
#define PCV_GRID_BITS 7
#endif
constexpr int kGridBits = PCV_GRID_BITS;  // 128^3 cells, 2 MiB: stays in L2

__global__ __launch_bounds__(256) void spec_depth_grid_kernel(const uint32_t* __restrict__ walk, uint8_t* __restrict__ grid,
                                                              uint4* __restrict__ zero, uint32_t zero_vecs) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;  // grid is exactly 2^21 cells
  // the exact per-leaf counters of the count that follows the chain pass are cleared here: a fill on the side stream would have
  // to be joined in front of that count (a cross-stream wait between two kernels is ~15 us of idle stream)
  for (uint32_t j = c; j < zero_vecs; j += 1u << (3 * kGridBits)) zero[j] = make_uint4(0u, 0u, 0u, 0u);
  constexpr uint32_t kM = (1u << kGridBits) - 1u;
  const uint32_t ix = c & kM, iy = (c >> kGridBits) & kM, iz = c >> (2 * kGridBits);
  uint32_t r = walk[0];
  uint32_t l = 0;
  while (!(r & PCV_SPEC_LEAF) && l < (uint32_t)kGridBits) {
    ++l;
    const uint32_t d = ((ix >> (kGridBits - l)) & 1u) << 2 | ((iy >> (kGridBits - l)) & 1u) << 1 | ((iz >> (kGridBits - l)) & 1u);
    r = walk[(r & PCV_SPEC_INDEX_MASK) + d];
  }
  grid[c] = (uint8_t)((r & PCV_SPEC_LEAF) ? l : kGridBits + 1);
}

__device__ __forceinline__ uint32_t pcv_wave_incl_scan32(uint32_t v) {  // inclusive prefix over lanes 0..31 (DPP rows 0 and 1)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
  return v;
}
__device__ __forceinline__ uint32_t pcv_lane_again() {  // the lane number, recomputed (opaque to the compiler: nothing to keep live)
  uint32_t zero = 0;
  asm volatile("" : "+v"(zero));
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
}

// ---- the single chain pass (round 5): the ONE chain kernel of the shipped library ---------------------------------------------
// Rounds 2-4 built this pass up (their three kernel generations were deleted in round 6; the A/B records stay under profiles/,
// the text in the git history of pcv_encode_exp.inc): depth-dealt tiles of two points
// per lane (a workgroup of BLOCK lanes takes 2 x BLOCK points through ONE memory phase, deals them into groups of 64 by
// predicted depth, wave w walks group w and then group 2 x waves - 1 - w), the deal by one returning LDS atomic per point,
// octant bits as lane masks, first-candidate codes behind a wave-uniform branch, 12-byte records that leave through LDS in
// input order, Float32 codes in the dense `wide` pool (one reservation per wave). Round 5 changes what a level step COSTS
// and what surrounds it (43 % of the issue cycles were not level arithmetic):
//   * CODES FROM CODES. A Float32-coded level re-decodes the position from a 24-bit code v, and the next level's code is,
//     exactly, w = 2 v - bit — a float again. The f64 chain only perturbs that by its rounding noise (<= 2^-40 of the cube
//     for the bench cloud), far below half a float ulp unless w is tiny: wherever PcvLevels::code_thr_hi admits the step
//     (pcv_make_levels proves the bound per level) the pass keeps w and the level costs one compare, one select and two
//     FMAs per coordinate instead of a subtraction, an exact division, a clamp, two conversions and the decode FMA —
//     3 of the 4 Float32-coded steps of a bench-cloud point. A wave that holds a code below the threshold (or exactly 1/2 or
//     1, where ties and clamps decide) runs the full step for that level; the position is materialised once, when the
//     Float32-coded levels end.
//   * RAW vs ROUTED input is a template parameter: the raw instantiation carries no level-1 decode, no per-lane selects of
//     the root cube, none of its registers.
//   * u16- and u8-coded levels share one loop (255 / 65535 and their reciprocals are scalars), so the walk has three loop
//     bodies instead of four + the cold per-level switch.
//   * The tame test is made once per point in the front phase (integer compare of the exponent words) and travels as a
//     bit of the dealt slot number.
//   * The record epilogue branches on a ballot of "Float32-coded record level": a wave runs one conversion arm; slots,
//     bounds and the pool region are 32-bit / scalar arithmetic; intensity bits are copied in input order by the closing
//     phase (coalesced) instead of by the dealt lanes.
// Same walk, same records, same bytes.
/* walk record of T'' node `idx`: the first TOP records (T'' is level-major: the top of the tree) are mirrored in LDS */
#define CP_WALK_AT(idx) cp_walk_at<TOP>(walk, swalk, (idx))
template <int TOP>
__device__ __forceinline__ uint32_t cp_walk_at(const uint32_t* __restrict__ walk, const uint32_t* swalk, uint32_t idx) {
  if (TOP == 0) return walk[idx];
  return idx < (uint32_t)TOP ? swalk[idx] : walk[idx];
}
#define CP_KEEP_STEP                                                                                                    \
  if (KEEP) {                                                                                                           \
    const bool take = (rec & PCV_SPEC_CANDIDATE) && kl == 0;                                                            \
    if (__builtin_amdgcn_ballot_w64(take) != 0ull) {                                                                    \
      asm volatile(""); /* keeps the copies inside the branch (the compiler would if-convert them into every level) */  \
      if (take) {                                                                                                       \
        kx = vx, ky = vy, kz = vz;                                                                                      \
        kl = U; /* candidates have level >= 1 */                                                                        \
      }                                                                                                                 \
    }                                                                                                                   \
  }
/* the octant bits of level U + 1 from level U's state: mode 1 / 2 = from its integer / Float32 codes (a Float32 code of
   exactly 1/2 is a tie of the exact values: the comparison against the centre decides), else the comparison itself */
#define CP_BITS(GUARD)                                                                                                  \
  PcvOctBits b;                                                                                                         \
  if (!GUARD && mode == 1u) {                                                                                           \
    b = pcv_bits_from_codes(half, vx, vy, vz);                                                                          \
  } else if (!GUARD && mode == 2u) {                                                                                    \
    b = pcv_bits_from_codes(0.5, vx, vy, vz);                                                                           \
    if (__builtin_expect(__any(pcv_f32_code_tie(vx, vy, vz)), 0)) b = pcv_chain_bits(lv.edge[U], px, py, pz, mx, my, mz); \
  } else {                                                                                                              \
    b = pcv_chain_bits(lv.edge[U], px, py, pz, mx, my, mz);                                                             \
  }
/* Levels U + 1 .. LEND of the lanes that have not reached a leaf. U is the wave's level counter (every live lane is at
   level U), so the level constants are scalars; L is the lane's own last level. The child's walk record is requested
   right after the digit and lands while the level's arithmetic runs. */
#define CP_LOOP(GUARD, LEND, SCALARS, APPLY)                                                                            \
  while (U < (LEND)) {                                                                                                  \
    const bool live = !(rec & PCV_SPEC_LEAF);                                                                           \
    if (!__any(live)) break;                                                                                            \
    const double half_next = lv.digit_half[U + 1];                                                                      \
    const uint32_t mode_next = lv.digit_mode[U + 1];                                                                    \
    SCALARS                                                                                                             \
    if (live) {                                                                                                         \
      CP_KEEP_STEP                                                                                                      \
      CP_BITS(GUARD)                                                                                                    \
      const uint32_t next = CP_WALK_AT((rec & PCV_SPEC_INDEX_MASK) + b.digit());                                        \
      const double ec = lv.edge[U + 1];                                                                                 \
      const PcvRecip ic{lv.inv_edge[U + 1], lv.inv_edge_lo[U + 1]};                                                     \
      APPLY;                                                                                                            \
      rec = next;                                                                                                       \
      L = U + 1;                                                                                                        \
    }                                                                                                                   \
    half = half_next;                                                                                                   \
    mode = mode_next;                                                                                                   \
    ++U;                                                                                                                \
  }
/* Float32 codes of level U + 1 from those of level U (see the header comment and pcv_make_levels). A wave with a code the
   table does not admit takes the full step for ALL its live lanes: it is exact everywhere. Positions are not kept. */
#define CP_CODE_LOOP(LEND)                                                                                              \
  while (U < (LEND)) {                                                                                                  \
    const bool live = !(rec & PCV_SPEC_LEAF);                                                                           \
    if (!__any(live)) break;                                                                                            \
    U = __builtin_amdgcn_readfirstlane(U);                                                                              \
    const uint32_t thr_hi = lv.code_thr_hi[U];                                                                          \
    if (live) {                                                                                                         \
      CP_KEEP_STEP                                                                                                      \
      const PcvOctBits b = pcv_bits_from_codes(0.5, vx, vy, vz);                                                        \
      const double fx = __hiloint2double(b.x ? 0x3ff00000 : 0, 0), fy = __hiloint2double(b.y ? 0x3ff00000 : 0, 0),      \
                   fz = __hiloint2double(b.z ? 0x3ff00000 : 0, 0);                                                      \
      const double wx = __fma_rn(2.0, vx, -fx), wy = __fma_rn(2.0, vy, -fy), wz = __fma_rn(2.0, vz, -fz); /* exact */  \
      const uint32_t hx = (uint32_t)__double2hiint(wx), hy = (uint32_t)__double2hiint(wy), hz = (uint32_t)__double2hiint(wz); \
      const uint32_t hmin = min(min(hx, hy), hz), hmax = max(max(hx, hy), hz);                                          \
      const double ec = lv.edge[U + 1];                                                                                 \
      uint32_t next;                                                                                                    \
      if (__builtin_expect(__any(hmin < thr_hi || hmax >= 0x3ff00000u), 0)) {                                           \
        double px = __fma_rn(vx, lv.edge[U], mx), py = __fma_rn(vy, lv.edge[U], my), pz = __fma_rn(vz, lv.edge[U], mz); \
        PcvOctBits bb = b;                                                                                              \
        if (__any(pcv_f32_code_tie(vx, vy, vz))) bb = pcv_chain_bits(lv.edge[U], px, py, pz, mx, my, mz);               \
        next = CP_WALK_AT((rec & PCV_SPEC_INDEX_MASK) + bb.digit());                                                    \
        const PcvRecip ic{lv.inv_edge[U + 1], lv.inv_edge_lo[U + 1]};                                                   \
        pcv_chain_apply_bits_t<PCV_ENC_FLOAT32, false>(bb, ec, ic, px, py, pz, mx, my, mz, vx, vy, vz);                 \
      } else {                                                                                                          \
        next = CP_WALK_AT((rec & PCV_SPEC_INDEX_MASK) + b.digit());                                                     \
        mx = __fma_rn(fx, ec, mx), my = __fma_rn(fy, ec, my), mz = __fma_rn(fz, ec, mz); /* pcv_step_min */             \
        vx = wx, vy = wy, vz = wz;                                                                                      \
      }                                                                                                                 \
      rec = next;                                                                                                       \
      L = U + 1;                                                                                                        \
    }                                                                                                                   \
    asm volatile("s_add_i32 %0, %0, 1" : "+s"(U) : : "scc"); /* ++U, kept scalar */                                     \
  }
/* the tame walk: [generic per-level switch: Float64-coded levels, non-monotone tables — cold] -> Float32-coded levels in
   full up to the first admitted code step -> code steps -> the position, once -> remaining Float32-coded levels -> integer-coded levels */
#define CP_WALK_TAME                                                                                                    \
  {                                                                                                                     \
    double half = lv.digit_half[U];                                                                                     \
    uint32_t mode = lv.digit_mode[U];                                                                                   \
    const int e0 = lv.first_f32 - 1 < lv.nlevels ? lv.first_f32 - 1 : lv.nlevels;                                       \
    const int e1 = lv.first_u16 - 1 < lv.nlevels ? lv.first_u16 - 1 : lv.nlevels;                                       \
    const int cb = lv.code_begin, ce = lv.code_end < lv.nlevels ? lv.code_end : lv.nlevels;                             \
    if (__builtin_expect(U < e0, 0)) {                                                                                  \
      CP_LOOP(false, e0, , pcv_chain_apply_bits<false>(lv.enc[U + 1], b, ec, ic, px, py, pz, mx, my, mz, vx, vy, vz))   \
    }                                                                                                                   \
    CP_LOOP(false, (e1 < cb ? e1 : cb), , (pcv_chain_apply_bits_t<PCV_ENC_FLOAT32, false>(b, ec, ic, px, py, pz, mx, my, mz, vx, vy, vz))) \
    if (U == cb && cb < ce) {                                                                                           \
      CP_CODE_LOOP(ce)                                                                                                  \
      U = __builtin_amdgcn_readfirstlane(U);                                                                            \
      px = __fma_rn(vx, lv.edge[U], mx), py = __fma_rn(vy, lv.edge[U], my), pz = __fma_rn(vz, lv.edge[U], mz);          \
      half = lv.digit_half[U];                                                                                          \
      mode = lv.digit_mode[U];                                                                                          \
      CP_LOOP(false, e1, , (pcv_chain_apply_bits_t<PCV_ENC_FLOAT32, false>(b, ec, ic, px, py, pz, mx, my, mz, vx, vy, vz))) \
    }                                                                                                                   \
    CP_LOOP(false, lv.nlevels,                                                                                          \
            const bool u8 = U + 1 >= lv.first_u8; const double maxval = u8 ? 255.0 : 65535.0;                           \
            const PcvRecip rmax = u8 ? PCV_RECIP_255 : PCV_RECIP_65535;,                                                \
            pcv_chain_apply_bits_int(b, ec, ic, maxval, rmax, px, py, pz, mx, my, mz, vx, vy, vz))                      \
  }
/* wild coordinates (NaN, infinities, |v| > 2^500) or an untamed table: every division guarded, every digit by comparison */
#define CP_WALK_GUARDED                                                                                                 \
  {                                                                                                                     \
    double half = 0.0;                                                                                                  \
    uint32_t mode = 0u;                                                                                                 \
    CP_LOOP(true, lv.nlevels, , pcv_chain_apply_bits<true>(lv.enc[U + 1], b, ec, ic, px, py, pz, mx, my, mz, vx, vy, vz)) \
  }

// DIAG (libpcv_hip_exp.so only). Bit 1 (PCV_COLOR_LATE=1, a correct build): the 12-byte records leave WITHOUT their colour — the
// record sort's first pass reads it from the caller's array as it loads the records (PcvSortPayload::color_in): no colour loads
// behind the walks (1.86 -> 1.73 ms for the pass at 100 M points; the sort pass loses more than that: not shipped).
template <bool KEEP, bool RAW, int BLOCK, int TOP = 0 /* walk records mirrored in LDS (a multiple of 4 x BLOCK, or 0) */, int DIAG = 0>
__global__ __launch_bounds__(BLOCK, 8) void chain_pass_kernel(
    PcvLevels lv, const uint32_t* __restrict__ walk, uint64_t n, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ z, PcvRouted routed, const uint8_t* __restrict__ color, uint32_t color_stride,
    const float* __restrict__ intensity, uint32_t* __restrict__ rank, uint4* __restrict__ payload, uint32_t* __restrict__ inten_bits,
    const uint8_t* __restrict__ depth_grid /* null: no deal by depth (small builds) */, float cells_per_unit /* 128 / root edge */,
    uint4* __restrict__ wide /* set: 12-byte records */, uint32_t* __restrict__ pool_ctr, uint32_t pool_cap) {
  constexpr int TILE = 2 * BLOCK, kGroups = TILE / 64;
  static_assert(TILE <= 1024, "a tile must not span two pool regions, and its slot numbers share 16 bits with the wild flag");
  constexpr uint32_t kWild = 0x8000u;
  __shared__ double sxyz[3 * TILE];  // slot s: half s / BLOCK, {x, y, z}[s % BLOCK]
  __shared__ uint16_t sidx[TILE];
  __shared__ uint32_t kcnt[32];  // points of the tile per depth class
  __shared__ uint32_t swalk[TOP ? TOP : 4];
  static_assert(TOP % 256 == 0, "whole waves of uint4 loads");
  constexpr int kTopRounds = (TOP + 4 * BLOCK - 1) / (4 * BLOCK);
  // the first half of the coordinates is dead once every wave has fetched its first group: it stages the tile's records
  uint32_t* const okey = reinterpret_cast<uint32_t*>(sxyz);    // TILE keys: the x of the first half
  uint2* const opay = reinterpret_cast<uint2*>(sxyz + BLOCK);  // TILE payloads: its y and z
  // the wave's number is a scalar and the lane number can be had again from nothing (mbcnt): no lane-indexed value needs to
  // stay in a register across the walks
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int lane = (int)pcv_lane_again(), tid = wave * 64 + lane;
  const uint64_t base = (uint64_t)blockIdx.x * TILE;
  const uint32_t here = n - base < (uint64_t)TILE ? (uint32_t)(n - base) : (uint32_t)TILE;  // points of this tile (scalar)
  const bool stage = wide != nullptr;  // grid-uniform: 12-byte records
  if (tid < 32) kcnt[tid] = 0;
  double qx[2], qy[2], qz[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t t = (uint32_t)(h * BLOCK + tid);
    qx[h] = qy[h] = qz[h] = 0.0;
    if (t < here) {
      if (RAW) {
        qx[h] = x[base + t], qy[h] = y[base + t], qz[h] = z[base + t];
      } else {  // routed input: the position the sending rank held after level 1 (decode of the level-1 codes)
        double t0, t1, t2, t3, t4, t5;
        uint32_t dd;
        (void)pcv_chain_start(lv, routed, x, y, z, base + t, qx[h], qy[h], qz[h], t0, t1, t2, t3, t4, t5, dd);
      }
    }
  }
  // the top of the walk table: requested behind the coordinates, stored to LDS before the deal's last barrier (the buffer is
  // allocated for at least 1 + 8 x 8 192 records: reading past the tree's last node is harmless, never indexed)
  uint4 wtop[TOP ? kTopRounds : 1];
  if (TOP) {
#pragma unroll
    for (int k = 0; k < kTopRounds; ++k)
      if ((k * BLOCK + tid) * 4 < TOP) wtop[k] = reinterpret_cast<const uint4*>(walk)[k * BLOCK + tid];  // (wave-uniform)
  }
  __syncthreads();  // the counters are zero (the coordinate loads are in flight)
  uint32_t key[2], pos[2], wild[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    key[h] = kSpecClasses - 1;  // padding lanes go last
    if ((uint32_t)(h * BLOCK + tid) < here) {
      key[h] = 0;
      if (depth_grid) {
        // cell of the 128^3 grid over the root cube (NaN -> 0, out of range clamps) -> predicted depth (1..8, 8 = deeper)
        constexpr float kTop = (float)((1 << kGridBits) - 1);
        const uint32_t ix = (uint32_t)fminf(fmaxf((float)(qx[h] - lv.root_min[0]) * cells_per_unit, 0.f), kTop);
        const uint32_t iy = (uint32_t)fminf(fmaxf((float)(qy[h] - lv.root_min[1]) * cells_per_unit, 0.f), kTop);
        const uint32_t iz = (uint32_t)fminf(fmaxf((float)(qz[h] - lv.root_min[2]) * cells_per_unit, 0.f), kTop);
        key[h] = (uint32_t)(kSpecClasses - 2) - depth_grid[ix | (iy << kGridBits) | (iz << (2 * kGridBits))];  // deepest first
      }
    }
    // pcv_point_is_tame, a little stricter (|v| < 2^500: one integer compare of the largest exponent word; NaN and the
    // infinities have the largest of all) — the guarded walk is exact for every input, so "wild" may be over-reported
    const uint32_t ex = max(max((uint32_t)__double2hiint(qx[h]) & 0x7fffffffu, (uint32_t)__double2hiint(qy[h]) & 0x7fffffffu),
                            (uint32_t)__double2hiint(qz[h]) & 0x7fffffffu);
    wild[h] = ex >= 0x5f300000u ? kWild : 0u;  // 2^500
  }
  // rank of a point among the tile's points of its class (any order inside a class will do: the deal only decides which
  // lane walks which point)
#pragma unroll
  for (int h = 0; h < 2; ++h) pos[h] = __hip_atomic_fetch_add(&kcnt[key[h]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();  // the counts are final
  const uint32_t cnt = kcnt[lane & 31];
  const uint32_t inc = pcv_wave_incl_scan32(lane < 32 ? cnt : 0u);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t slot = (uint32_t)__shfl((int)(inc - cnt), (int)key[h], 64) + pos[h];  // classes ascending = deepest first
    if (RAW) {
      double* const c = sxyz + (slot / BLOCK) * (3 * BLOCK) + (slot % BLOCK);
      c[0] = qx[h], c[BLOCK] = qy[h], c[2 * BLOCK] = qz[h];
    }
    sidx[slot] = (uint16_t)((uint32_t)(h * BLOCK + tid) | wild[h]);
  }
  if (TOP) {
    tid = wave * 64 + (int)pcv_lane_again();
#pragma unroll
    for (int k = 0; k < kTopRounds; ++k)
      if ((k * BLOCK + tid) * 4 < TOP) reinterpret_cast<uint4*>(swalk)[k * BLOCK + tid] = wtop[k];
  }
  __syncthreads();
  const uint32_t rec0 = walk[0];  // the root's record (scalar)
  const bool monotone = lv.first_f32 < (1 << 20);
#pragma unroll 1
  for (int task = 0; task < 2; ++task) {
    lane = (int)pcv_lane_again();
    const int s = (task == 0 ? wave : kGroups - 1 - wave) * 64 + lane;  // first the deep end, then the shallow end
    const uint32_t jd = sidx[s];
    double px = 0, py = 0, pz = 0;
    if (RAW) {
      const double* const c = sxyz + (s / BLOCK) * (3 * BLOCK) + (s % BLOCK);
      px = c[0], py = c[BLOCK], pz = c[2 * BLOCK];
    }
    if (task == 0 && stage) __syncthreads();  // the first half now belongs to the records (the waves are still in step here)
    const uint32_t j = jd & (kWild - 1u);
    if (j < here) {
      double mx = lv.root_min[0], my = lv.root_min[1], mz = lv.root_min[2];
      double vx = 0, vy = 0, vz = 0;
      double kx = 0, ky = 0, kz = 0;
      int kl = 0;
      int L = 0;
      uint32_t rec = rec0;
      if (!RAW) {
        uint32_t d1 = 0;
        if (pcv_chain_start(lv, routed, x, y, z, base + j, px, py, pz, mx, my, mz, vx, vy, vz, d1) == 2 && !(rec & PCV_SPEC_LEAF)) {
          L = 1;  // level 1 is given (digit + codes)
          rec = walk[(rec & PCV_SPEC_INDEX_MASK) + d1];
        }
      }
      // the wave's level counter (all lanes start at the same level: 0, or 1 for routed input). readfirstlane pins it to a
      // scalar register: the level constants are then scalar loads and the loop control is SALU
      // (made opaque first: a constant start would be folded and the compiler then keeps the counter in a VGPR)
      int Ls = L;
      asm volatile("" : "+v"(Ls));
      int U = __builtin_amdgcn_readfirstlane(Ls);
      if (lv.fast_ok && !(jd & kWild)) {
        CP_WALK_TAME
      } else {
        CP_WALK_GUARDED
      }
      // the record carries the codes of the first candidate on the path where there is one, else those of the predicted leaf
      if (KEEP && kl) {
        vx = kx, vy = ky, vz = kz;
        L = kl;
      }
      // is the record's level Float32-coded (-> a `wide` pool entry)? The table is monotone (Float64 -> Float32 -> u16 -> u8 with
      // depth) unless first_f32 says "never": two compares instead of a gather from the level table at the very end of the
      // wave's critical path
      const bool is_wide = monotone ? (L < lv.first_u16 && (L >= lv.first_f32 || lv.enc[0] > PCV_ENC_UINT16)) : lv.enc[L] > PCV_ENC_UINT16;
      // the slot number again, from the dealt slot: nothing lane-indexed stays live across the loops
      uint32_t jj = j;
      asm volatile("" : "+v"(jj));
      if (stage) {
        const uint32_t key = (rec & PCV_SPEC_INDEX_MASK) << 8;  // the blue byte joins in the closing phase
        const uint64_t wm = __ballot(is_wide);
        uint2 out;
        if (wm == 0ull) {  // wave-uniform: integer codes only (the deep groups)
          out = make_uint2((uint32_t)vx | ((uint32_t)vy << 16), (uint32_t)vz);
        } else {
          // value domain -> raw code: the IEEE bits of the float (single-chain builds have no Float64-coded level:
          // build_begin_impl sends those to the exact pipeline)
          const uint32_t ccx = is_wide ? __float_as_uint((float)vx) : (uint32_t)vx, ccy = is_wide ? __float_as_uint((float)vy) : (uint32_t)vy,
                         ccz = is_wide ? __float_as_uint((float)vz) : (uint32_t)vz;
          out = make_uint2(ccx | (ccy << 16), ccz);
          // ONE reservation per wave in the pool region of the tile's slice of 1 024 input points (scalar: a tile never
          // spans two slices); the order in which waves reach the counter does not matter — the record names its entry
          const uint32_t region = (uint32_t)(base >> 10) & (kPcvPoolRegions - 1u);
          const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
          uint32_t first = 0;
          if (is_wide && below == 0u)
            first = __hip_atomic_fetch_add(pool_ctr + region, (uint32_t)__popcll(wm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          first = region * pool_cap + (uint32_t)__builtin_amdgcn_readlane((int)first, (int)__builtin_ctzll(wm));
          if (is_wide) {
            const uint32_t e = first + below;
            wide[e] = make_uint4(ccx, ccy, ccz, 0u);
            out = make_uint2(e, 0u);
          }
        }
        okey[jj] = key;
        opay[jj] = out;
      } else {  // 20-byte records (a predicted tree that could outgrow 24 rank bits): the codes travel in full
        const uint64_t i = base + jj;
        const uint32_t rgb = pcv_load_rgb(color + i * color_stride, i + 1 < n);
        rank[i] = rec & PCV_SPEC_INDEX_MASK;
        payload[i] = is_wide ? make_uint4(__float_as_uint((float)vx), __float_as_uint((float)vy), __float_as_uint((float)vz), rgb)
                             : make_uint4((uint32_t)vx, (uint32_t)vy, (uint32_t)vz, rgb);
      }
    }
  }
  // closing phase, input order again: full lines; the colour and the intensity bits join here
  tid = wave * 64 + (int)pcv_lane_again();
  if (stage) {
    uint32_t rgb[2];
    // (round 6, measured: requesting the colour in the front phase and parking it in LDS, or fetching it as aligned dwords + two
    // ds_bpermute per lane, changes nothing — 1.85 ms either way against 1.72 without any colour (a timing-only variant): what the
    // colour costs the pass is neither the latency in front of this barrier nor the unaligned loads;
    // profiles/r06_ab_colour_joins_in_the_sort_dropped.json)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t t = (uint32_t)(h * BLOCK + tid);
      rgb[h] = (!(DIAG & 1) && t < here) ? pcv_load_rgb(color + (base + t) * color_stride, base + t + 1 < n) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t t = (uint32_t)(h * BLOCK + tid);
      if (t < here) {
        const uint2 q = opay[t];
        // streaming stores: the records are read next by other kernels, long after they have left the L2 — written through, they
        // leave less for the write-back at the kernel's end (the pass and the count behind it: -15 us together, A B A B in one call)
        __builtin_nontemporal_store(okey[t] | (rgb[h] >> 16), &rank[base + t]);
        __builtin_nontemporal_store(((uint64_t)(q.y | ((rgb[h] & 0xffffu) << 16)) << 32) | q.x, reinterpret_cast<uint64_t*>(payload) + base + t);
      }
    }
  }
  if (inten_bits) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t t = (uint32_t)(h * BLOCK + tid);
      if (t < here) inten_bits[base + t] = __float_as_uint(intensity[base + t]);
    }
  }
}
#undef CP_WALK_GUARDED
#undef CP_WALK_TAME
#undef CP_CODE_LOOP
#undef CP_LOOP
#undef CP_BITS
#undef CP_KEEP_STEP
#undef CP_WALK_AT


// Exact number of points per predicted leaf: LDS-privatised histogram of the rank array over the bins
// [bin_base, bin_base + nbins), nbins <= kHistBins; one flush of the non-zero bins per workgroup.
constexpr int kHistBins = 16384;      // 64 KiB of LDS: the plain count (512 workgroups, two to a CU)
constexpr int kHistBinsRows = 32768;  // 128 KiB: the count over the record sort's workgroups with the rows kept (one to a CU)
__global__ __launch_bounds__(1024) void rank_hist_kernel(const uint32_t* __restrict__ rank, uint64_t n, uint64_t chunk,
                                                          uint32_t bin_base, uint32_t nbins, uint32_t* __restrict__ counts,
                                                          int shift /* 8: 12-byte records, the rank sits above the blue byte */,
                                                          uint32_t* __restrict__ rows /* set: this workgroup's histogram is kept,
                                                                                         rows[blockIdx.x * row_stride + bin_base + b] */,
                                                          uint32_t row_stride) {
  extern __shared__ uint32_t hist[];  // nbins counters
  for (uint32_t b = threadIdx.x; b < nbins; b += 1024) hist[b] = 0;
  __syncthreads();
  const uint64_t begin = (uint64_t)blockIdx.x * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;
  // chunk is a multiple of 4 and the array comes from the pool (256-byte aligned): 16-byte loads
  for (uint64_t i = begin + (uint64_t)threadIdx.x * 4; i < end; i += 4096) {
    if (i + 4 <= end) {
      const uint4 v = *reinterpret_cast<const uint4*>(rank + i);
      const uint32_t r[4] = {(v.x >> shift) - bin_base, (v.y >> shift) - bin_base, (v.z >> shift) - bin_base,
                             (v.w >> shift) - bin_base};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r[k] < nbins) atomicAdd(&hist[r[k]], 1u);
    } else {
      for (uint64_t j = i; j < end; ++j) {
        const uint32_t r = (rank[j] >> shift) - bin_base;
        if (r < nbins) atomicAdd(&hist[r], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nbins; b += 1024) {
    const uint32_t v = hist[b];
    if (v) atomicAdd(&counts[bin_base + b], v);
    if (rows) rows[(uint64_t)blockIdx.x * row_stride + bin_base + b] = v;
  }
}

// Chain continuation over the sorted slots of the true leaves that lie BELOW the first candidate node of their path (the
// candidate was split): their records carry the codes of the candidate's level, and a point's chain from that level on
// is a function of those codes alone — decode in the candidate's cube, then the ordinary level steps down to the leaf
// (the digits come out of the chain itself and are the ones the chain pass walked with). One workgroup per <= kPcvSettleTile (1 024) slots
// of one leaf; range and levels are wave-uniform, so the level loop and its encoding switch are scalar.
struct alignas(16) PcvContRange {
  uint32_t from_level, to_level, pad0, pad1;
  double mn[3];  // cube min of the candidate node (NodeId::find_bounding_cube recurrence, as the node records carry it)
  double pad2;
};
static_assert(sizeof(PcvContRange) == 48, "continue range");
// raw codes of level rg.from_level in the candidate's cube -> raw codes of level rg.to_level down the point's own path
__device__ __forceinline__ void pcv_continue_codes(const PcvLevels& lv, const PcvContRange& rg, uint32_t& cx, uint32_t& cy, uint32_t& cz) {
  const int from = (int)rg.from_level, to = (int)rg.to_level;
  const uint32_t fe = lv.enc[from], te = lv.enc[to];
  double mx = rg.mn[0], my = rg.mn[1], mz = rg.mn[2];
  const double e = lv.edge[from];
  double px = pcv_decode_coord(fe, cx, mx, e), py = pcv_decode_coord(fe, cy, my, e), pz = pcv_decode_coord(fe, cz, mz, e);
  double vx = 0, vy = 0, vz = 0;
  for (int L = from + 1; L <= to; ++L)
    (void)pcv_chain_level<true>(lv.enc[L], lv.edge[L - 1], lv.edge[L], PcvRecip{lv.inv_edge[L], lv.inv_edge_lo[L]}, px, py, pz, mx, my,
                                mz, vx, vy, vz);
  cx = (uint32_t)pcv_val_to_code(te, vx);
  cy = (uint32_t)pcv_val_to_code(te, vy);
  cz = (uint32_t)pcv_val_to_code(te, vz);
}
// Stand-alone form (the slot-wise settle kernel and the 20-byte records of the exact geometry use it; the leaf-wise settle
// kernel continues the chain itself, see promote_settle_leaf_kernel): rewrites the codes in place.
template <bool kCompact>
__global__ __launch_bounds__(256) void spec_continue_kernel(PcvLevels lv, const PcvContRange* __restrict__ ranges,
                                                             const PcvSettleItem* __restrict__ items, uint4* __restrict__ payload,
                                                             uint4* __restrict__ wide /* kCompact: codes of Float32-coded levels */) {
  const PcvSettleItem it = items[blockIdx.x];
  const PcvContRange rg = ranges[it.rank];
  const uint32_t fe = lv.enc[rg.from_level], te = lv.enc[rg.to_level];
  uint2* __restrict__ pay2 = reinterpret_cast<uint2*>(payload);
  for (uint32_t s = it.begin + threadIdx.x; s < it.end; s += 256) {
    uint4 p = make_uint4(0, 0, 0, 0);
    uint2 q = make_uint2(0, 0);
    uint32_t idx = 0;
    if (kCompact) {
      q = pay2[s];
      if (fe <= PCV_ENC_UINT16) {
        p.x = q.x & 0xffffu, p.y = q.x >> 16, p.z = q.y & 0xffffu;
      } else {
        idx = q.x;
        const uint4 w = wide[idx];
        p.x = w.x, p.y = w.y, p.z = w.z;
      }
    } else {
      p = payload[s];
    }
    pcv_continue_codes(lv, rg, p.x, p.y, p.z);
    if (!kCompact) {
      payload[s] = p;
    } else if (te <= PCV_ENC_UINT16) {
      pay2[s] = make_uint2(p.x | (p.y << 16), (q.y & 0xffff0000u) | p.z);
    } else {  // encodings narrow with depth: a Float32-coded leaf level means the candidate's level was Float32-coded too,
              // so the record already carries the input index
      wide[idx] = make_uint4(p.x, p.y, p.z, 0u);
    }
  }
}

// Chain replay over the sorted slots of the flagged leaves. ranges[k] = {first slot, slots before this range in the
// flattened work list, level}; `total` = all flagged slots.
struct PcvFixRange {
  uint32_t lo, before, level, pad;
};
__global__ __launch_bounds__(256) void spec_replay_kernel(PcvLevels lv, const PcvFixRange* __restrict__ ranges, uint32_t num_ranges,
                                                           uint32_t total, const double* __restrict__ x,
                                                           const double* __restrict__ y, const double* __restrict__ z,
                                                           PcvRouted routed, uint4* __restrict__ payload,
                                                           uint4* __restrict__ wide /* set: 12-byte records */,
                                                           uint32_t pool_cap /* entries per pool region */) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
    uint32_t a = 0, b = num_ranges;  // last range with before <= j
    while (b - a > 1) {
      const uint32_t mid = (a + b) >> 1;
      if (ranges[mid].before <= j) a = mid;
      else b = mid;
    }
    const PcvFixRange rg = ranges[a];
    const uint64_t s = (uint64_t)rg.lo + (j - rg.before);
    uint4 p = make_uint4(0, 0, 0, 0);
    uint2 q = make_uint2(0, 0);
    if (wide) {
      q = reinterpret_cast<const uint2*>(payload)[s];
      p.x = q.x;
    } else {
      p = payload[s];
    }
    const uint64_t i = p.x;  // the input index the finalize kernel left here
    const int target = (int)rg.level;
    double px, py, pz, mx, my, mz;
    double vx = 0, vy = 0, vz = 0;
    uint32_t d1;
    int L = pcv_chain_start(lv, routed, x, y, z, i, px, py, pz, mx, my, mz, vx, vy, vz, d1) - 1;
    while (L < target) {
      ++L;
      (void)pcv_chain_level<true>(lv.enc[L], lv.edge[L - 1], lv.edge[L], PcvRecip{lv.inv_edge[L], lv.inv_edge_lo[L]}, px, py, pz, mx,
                                  my, mz, vx, vy, vz);
    }
    const uint32_t en = lv.enc[target];
    p.x = (uint32_t)pcv_val_to_code(en, vx);
    p.y = (uint32_t)pcv_val_to_code(en, vy);
    p.z = (uint32_t)pcv_val_to_code(en, vz);
    if (!wide) {
      payload[s] = p;
    } else if (en <= PCV_ENC_UINT16) {
      reinterpret_cast<uint2*>(payload)[s] = make_uint2(p.x | (p.y << 16), (q.y & 0xffff0000u) | p.z);
    } else {  // the codes go where `settle` looks for those of a Float32-coded leaf: a pool entry, named by the record. The
              // chain pass fills every pool region from its bottom; the j-th replayed slot takes an entry from the TOP of
              // region j % regions — the host has checked that the two cannot meet (single_chain_topology)
      const uint32_t e = (j & (kPcvPoolRegions - 1u)) * pool_cap + pool_cap - 1u - j / kPcvPoolRegions;
      wide[e] = make_uint4(p.x, p.y, p.z, 0u);
      reinterpret_cast<uint2*>(payload)[s] = make_uint2(e, q.y);
    }
  }
}

// K6 runs as two kernels over the sorted records. Seven of eight points stay in their leaf: `settle` streams over all
// slots (two per lane, both record chains started before either is consumed) and finishes those with straight-line
// code. The every-8th points climb a data-dependent number of levels (decode + encode per level): `settle` copies
// their records into a COMPACT array (climber k of leaf r sits at climb_base[r] + k, so the array is dense and in slot
// order) and `climb` runs one lane per entry — its waves are full of climbers and it reads 32 bytes per climber instead
// of touching every sector of the 16-byte payload array to use an eighth of it.
#ifdef PCV_EXPERIMENTS
// (libpcv_hip_exp.so, PCV_WIDE_MASK=m) the wide-code gathers of `settle` fold onto the first m + 1 entries: WRONG output,
// it only times the kernel without the line fetches the sparse gather costs (the bound of any fix for them)
__constant__ uint32_t pcv_exp_wide_mask = 0xffffffffu;
// (PCV_SETTLE_XCD=1) number of settle items when workgroup b takes item (b % 8) * ceil(items / 8) + b / 8: workgroups go
// round-robin over the 8 XCDs, so consecutive items (slices of one leaf, neighbouring leaves) then share one L2
__constant__ uint32_t pcv_exp_settle_items = 0;
#define PCV_WIDE_INDEX(i) ((i) & pcv_exp_wide_mask)
#else
#define PCV_WIDE_INDEX(i) (i)
#endif
// (PcvClimber: pcv_settle_dev.h)

// one sorted slot of `settle`: finish it in its leaf, or hand it to `climb`
// kClimb16: the climber record is the 16-byte payload alone {code x, code y, code z, rgb} — leaf, slot and position in the
// parent follow from the record's index (climbers are dense per leaf, in slot order: index = climb_base[leaf] + j / 8);
// used by the leaf-wise kernels when there is no intensity and no Float64 level (half the climber traffic)
template <bool kCompact, bool kClimb16 = false>
__device__ __forceinline__ void settle_one(const PcvPromoteTables& pt, uint64_t s, const PcvNodeRec& c, uint32_t r, uint4 p,
                                           const uint32_t h[3], uint32_t inten, const uint32_t* __restrict__ climb_base,
                                           PcvClimber* __restrict__ climbers, const PromoteOut& o,
                                           const uint4* __restrict__ wide) {
  if (kCompact) {
    if (c.enc <= PCV_ENC_UINT16) {
      p.y = p.x >> 16;
      p.x &= 0xffffu;
    } else {  // Float32-coded leaf: p.x is the input index
      const uint4 w = wide[PCV_WIDE_INDEX(p.x)];
      p.x = w.x, p.y = w.y, p.z = w.z;
    }
  }
  const uint32_t j = (uint32_t)s - c.lo;
  if (c.parent == 0xffffffffu || (j & 7u) != 0) promote_one<false>(pt, s, c, p, h[0], h[1], h[2], inten, o);
  else if (kClimb16) reinterpret_cast<uint4*>(climbers)[climb_base[r] + (j >> 3)] = p;
  else climbers[climb_base[r] + (j >> 3)] = PcvClimber{p, r, (uint32_t)s, inten, 0u};
}

// kSettleSlots sorted slots per lane: every record load of the tile is in flight before the first is used
// kCompact: 12-byte records (key = rank << 8 | blue, uint2 payload); `wide` holds the codes of the points whose leaf
// level is Float32-coded, by input index (the record's first word).
template <int kSettleSlots, bool kCompact>
__global__ __launch_bounds__(256) void promote_settle_kernel(
    PcvPromoteTables pt, uint64_t n, const uint32_t* __restrict__ rank, const uint4* __restrict__ payload,
    const uint32_t* __restrict__ cx_hi, const uint32_t* __restrict__ cy_hi, const uint32_t* __restrict__ cz_hi,
    const uint32_t* __restrict__ inten_bits, const uint32_t* __restrict__ climb_base, PcvClimber* __restrict__ climbers,
    PromoteOut o, const uint4* __restrict__ wide) {
  const uint64_t base = (uint64_t)blockIdx.x * (256 * kSettleSlots) + threadIdx.x;
  uint32_t r[kSettleSlots];
  uint4 p[kSettleSlots];
  uint32_t h[kSettleSlots][3], in[kSettleSlots];
#pragma unroll
  for (int k = 0; k < kSettleSlots; ++k) {
    const uint64_t s = base + 256 * k;
    const bool live = s < n;
    r[k] = live ? rank[s] : 0u;
    if (kCompact) {
      const uint2 q = live ? reinterpret_cast<const uint2*>(payload)[s] : make_uint2(0, 0);
      p[k] = make_uint4(q.x, 0u, q.y & 0xffffu, (q.y >> 16) | ((r[k] & 0xffu) << 16));  // x: both codes, or the input index
      r[k] >>= 8;
    } else {
      p[k] = live ? payload[s] : make_uint4(0, 0, 0, 0);
    }
    h[k][0] = h[k][1] = h[k][2] = 0;
    in[k] = 0;
    if (cx_hi && live) {
      h[k][0] = cx_hi[s];
      h[k][1] = cy_hi[s];
      h[k][2] = cz_hi[s];
    }
    if (inten_bits && live) in[k] = inten_bits[s];
  }
#pragma unroll
  for (int k = 0; k < kSettleSlots; ++k) {
    const uint64_t s = base + 256 * k;
    const bool live = s < n;
    if (!__any(live)) break;
    // 64 consecutive sorted slots almost always lie in one leaf (a leaf of the bench cloud spans 250 waves): its
    // 80-byte record then comes through the scalar cache into SGPRs instead of being gathered into 20 VGPRs per lane
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)r[k]);
    if (__all(!live || r[k] == u)) {
      if (live) settle_one<kCompact>(pt, s, pt.leaf_rec[u], r[k], p[k], h[k], in[k], climb_base, climbers, o, wide);
    } else if (live) {
      settle_one<kCompact>(pt, s, pt.leaf_rec[r[k]], r[k], p[k], h[k], in[k], climb_base, climbers, o, wide);
    }
  }
}

// Leaf-wise settle: one workgroup per work item (<= 512 consecutive slots of one leaf). The item and the leaf's record
// are wave-uniform scalar loads that do not depend on the records, so they run beside the record loads (in the slot-wise
// kernel above the leaf is only known once the rank has arrived: load -> readfirstlane -> scalar load, a wave lived
// 7 us, three quarters of it waiting); no divergent "two leaves in one wave" path either. The rank array is only read
// for the blue byte of the packed records.
// A leaf below a split first candidate (pcv_spec.h) arrives with the candidate level's codes: its items carry the index of
// the leaf's continuation range (PcvSettleItem::pad = 1 + index), and the workgroup continues the chain to the leaf's
// level in registers before it settles the slot — no separate pass over those records, no launch.
template <bool kCompact, bool kClimb16>
__global__ __launch_bounds__(256) void promote_settle_leaf_kernel(
    PcvPromoteTables pt, const PcvSettleItem* __restrict__ items, const uint32_t* __restrict__ rank,
    const uint4* __restrict__ payload, const uint32_t* __restrict__ cx_hi, const uint32_t* __restrict__ cy_hi,
    const uint32_t* __restrict__ cz_hi, const uint32_t* __restrict__ inten_bits, const uint32_t* __restrict__ climb_base,
    PcvClimber* __restrict__ climbers, PromoteOut o, const uint4* __restrict__ wide, PcvLevels lv,
    const PcvContRange* __restrict__ cont_ranges) {
  constexpr int kSlots = (int)kPcvSettleTile / 256;
  uint32_t item = blockIdx.x;
#ifdef PCV_EXPERIMENTS
  if (pcv_exp_settle_items) {
    item = (blockIdx.x & 7u) * ((pcv_exp_settle_items + 7u) / 8u) + (blockIdx.x >> 3);
    if (item >= pcv_exp_settle_items) return;
  }
#endif
  const PcvSettleItem it = items[item];
  // every record load is issued before anything is consumed; dead lanes of the leaf's last tile re-read the tile's
  // first slot (an item is never empty) so that no load sits behind a branch
  uint32_t key[kSlots], h[kSlots][3], in[kSlots];
  uint2 q[kSlots];
  uint4 p[kSlots];
#pragma unroll
  for (int k = 0; k < kSlots; ++k) {
    const uint32_t s = it.begin + threadIdx.x + 256 * k;
    const uint32_t sl = s < it.end ? s : it.begin;
    key[k] = 0;
    q[k] = make_uint2(0, 0);
    p[k] = make_uint4(0, 0, 0, 0);
    if (kCompact) {
      key[k] = rank[sl];
      q[k] = reinterpret_cast<const uint2*>(payload)[sl];
    } else {
      p[k] = payload[sl];
    }
    h[k][0] = h[k][1] = h[k][2] = 0;
    in[k] = 0;
    if (cx_hi) {
      h[k][0] = cx_hi[sl];
      h[k][1] = cy_hi[sl];
      h[k][2] = cz_hi[sl];
    }
    if (inten_bits) in[k] = inten_bits[sl];
  }
  const PcvNodeRec c = pt.leaf_rec[it.rank];
  if (it.pad != 0 && cont_ranges) {  // workgroup-uniform: this leaf continues its chain first
    const PcvContRange rg = cont_ranges[it.pad - 1];
    const uint32_t fe = lv.enc[rg.from_level];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
      const uint32_t s = it.begin + threadIdx.x + 256 * k;
      if (s >= it.end) continue;
      uint4 u = p[k];  // codes of the candidate's level, unpacked with ITS encoding
      if (kCompact) {
        u.w = (q[k].y >> 16) | ((key[k] & 0xffu) << 16);
        if (fe <= PCV_ENC_UINT16) {
          u.x = q[k].x & 0xffffu, u.y = q[k].x >> 16, u.z = q[k].y & 0xffffu;
        } else {
          const uint4 w = wide[PCV_WIDE_INDEX(q[k].x)];
          u.x = w.x, u.y = w.y, u.z = w.z;
        }
      }
      pcv_continue_codes(lv, rg, u.x, u.y, u.z);
      settle_one<false, kClimb16>(pt, s, c, it.rank, u, h[k], in[k], climb_base, climbers, o, nullptr);  // codes are unpacked already
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < kSlots; ++k) {
    const uint32_t s = it.begin + threadIdx.x + 256 * k;
    if (kCompact)  // x: both codes, or the input index
      p[k] = make_uint4(q[k].x, 0u, q[k].y & 0xffffu, (q[k].y >> 16) | ((key[k] & 0xffu) << 16));
    if (s < it.end) settle_one<kCompact, kClimb16>(pt, s, c, it.rank, p[k], h[k], in[k], climb_base, climbers, o, wide);
  }
}

__global__ __launch_bounds__(256) void promote_climb_kernel(
    PcvPromoteTables pt, uint32_t num_climbers, const PcvClimber* __restrict__ climbers, const uint32_t* __restrict__ cx_hi,
    const uint32_t* __restrict__ cy_hi, const uint32_t* __restrict__ cz_hi, PromoteOut o) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= num_climbers) return;
  const PcvClimber c = climbers[k];
  const PcvNodeRec rec = pt.leaf_rec[c.rank];
  uint32_t h[3] = {0, 0, 0};
  if (cx_hi) {
    h[0] = cx_hi[c.slot];
    h[1] = cy_hi[c.slot];
    h[2] = cz_hi[c.slot];
  }
  promote_one<true>(pt, c.slot, rec, c.pay, h[0], h[1], h[2], c.inten, o);
}

// Leaf-wise climb: one workgroup per <= 256 consecutive climber records of ONE leaf (they are dense per leaf). Every
// climber climbs at least once, from its leaf into the leaf's parent: both records are wave-uniform scalar loads that
// run beside the climber loads, and that first step is straight-line code.
// kClimb16: 16-byte climber records (the payload alone); the item's pad holds the leaf's first climber index, so the
// record's own index gives its slot (lo + 8 q) and its place in the parent's stream (child_off + q).
// Round 6: the climbers that go on past the parent — every eighth of them, 8 lanes of each wave, then 1 — are gathered through
// LDS into the first lanes of the workgroup and climb on there, with per-lane parent records: the per-lane loop costs one wave
// what it cost four (the kernel is compute-bound: ~300 VALU per climber for the first step and its final rewrite alone;
// profiles/r06_ab_climb_persistent_workgroups_dropped.json).
struct ClimbOn {
  uint64_t code[3];
  uint32_t rgb, inten, j, node;  // position j in the stream of node `node`
};
template <bool kClimb16>
__global__ __launch_bounds__(256) void promote_climb_leaf_kernel(PcvPromoteTables pt, const PcvSettleItem* __restrict__ items,
                                                                  const PcvClimber* __restrict__ climbers,
                                                                  const uint32_t* __restrict__ cx_hi,
                                                                  const uint32_t* __restrict__ cy_hi,
                                                                  const uint32_t* __restrict__ cz_hi, PromoteOut o) {
  __shared__ uint32_t wave_on[4];
  __shared__ ClimbOn queue[36];  // <= 256 / 8 + 1 climbers of an item go on
  const PcvSettleItem it = items[blockIdx.x];
  const uint32_t k = it.begin + threadIdx.x;
  const bool live = k < it.end;
  const uint32_t kk = live ? k : it.begin;
  uint4 pay;
  uint32_t slot_rel, inten = 0;
  const PcvNodeRec leaf = pt.leaf_rec[it.rank];
  if (kClimb16) {
    pay = reinterpret_cast<const uint4*>(climbers)[kk];
    slot_rel = (kk - it.pad) << 3;  // j of the climber inside its leaf's stream
  } else {
    const PcvClimber c = climbers[kk];
    pay = c.pay;
    slot_rel = c.slot - leaf.lo;
    inten = c.inten;
  }
  uint32_t h[3] = {0, 0, 0};
  if (!kClimb16 && cx_hi) {
    h[0] = cx_hi[leaf.lo + slot_rel];
    h[1] = cy_hi[leaf.lo + slot_rel];
    h[2] = cz_hi[leaf.lo + slot_rel];
  }
  const PcvNodeRec par = pt.node_rec[leaf.parent];  // a leaf with climbers is not the root
  uint64_t code[3] = {pay.x | ((uint64_t)h[0] << 32), pay.y | ((uint64_t)h[1] << 32), pay.z | ((uint64_t)h[2] << 32)};
  if (par.enc <= PCV_ENC_UINT16 && par.inv_edge != 0.0) {  // wave-uniform
    // integer codes in a tame cube (the host zeroes inv_edge otherwise): the decoded position lies inside the leaf's cube, which
    // lies inside the parent's — the exact constant-divisor division needs no range check (as in promote_final)
    const double maxval = par.enc == PCV_ENC_UINT8 ? 255.0 : 65535.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      code[a] = pcv_fix_encode<false>(pcv_decode_coord(leaf.enc, code[a], leaf.mn[a], leaf.edge), par.mn[a], par.edge,
                                      PcvRecip{par.inv_edge, par.inv_edge_lo}, maxval);
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double q = pcv_decode_coord(leaf.enc, code[a], leaf.mn[a], leaf.edge);
      code[a] = pcv_encode_coord(par.enc, q, par.mn[a], par.edge, PcvRecip{par.inv_edge, par.inv_edge_lo});
    }
  }
  const uint32_t j = leaf.child_off + (slot_rel >> 3);  // position in the parent's stream
  const bool goes_on = live && par.parent != 0xffffffffu && (j & 7u) == 0;
  if (live && !goes_on)  // stays in the parent: the final rewrite there
    promote_one<false>(pt, (uint64_t)par.lo + j, par, make_uint4((uint32_t)code[0], (uint32_t)code[1], (uint32_t)code[2], pay.w),
                       (uint32_t)(code[0] >> 32), (uint32_t)(code[1] >> 32), (uint32_t)(code[2] >> 32), inten, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t m = __ballot(goes_on);
  if (lane == 0) wave_on[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    before += w < wave ? wave_on[w] : 0u;
    total += wave_on[w];
  }
  if (total == 0) return;  // workgroup-uniform
  if (goes_on) {
    ClimbOn& q = queue[before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))];
    q.code[0] = code[0], q.code[1] = code[1], q.code[2] = code[2];
    q.rgb = pay.w, q.inten = inten, q.j = j, q.node = leaf.parent;
  }
  __syncthreads();
  if (threadIdx.x < total) {
    const ClimbOn q = queue[threadIdx.x];
    const PcvNodeRec cur = pt.node_rec[q.node];
    promote_one<true>(pt, (uint64_t)cur.lo + q.j, cur, make_uint4((uint32_t)q.code[0], (uint32_t)q.code[1], (uint32_t)q.code[2], q.rgb),
                      (uint32_t)(q.code[0] >> 32), (uint32_t)(q.code[1] >> 32), (uint32_t)(q.code[2] >> 32), q.inten, o);
  }
}

}  // namespace

void pcv_launch_leaf_encode(pcv_ctx* ctx, const PcvLevels& lv, const PcvWalkTables& wt, uint64_t n, const double* x,
                            const double* y, const double* z, const PcvRouted& routed, const uint8_t* color,
                            uint32_t color_stride, const float* intensity, uint32_t* rank, void* payload, uint32_t* cx_hi,
                            uint32_t* cy_hi, uint32_t* cz_hi, uint32_t* inten_bits) {
  if (n == 0) return;
  PcvProf prof(ctx, PCV_K_LEAF_ENCODE);
  hipLaunchKernelGGL(leaf_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lv, wt.walk, n, x,
                     y, z, routed, color, color_stride, intensity, rank, (uint4*)payload, cx_hi, cy_hi, cz_hi, inten_bits);
}

void pcv_launch_spec_encode(pcv_ctx* ctx, const PcvLevels& lv, const uint32_t* walk, uint64_t n, const double* x,
                            const double* y, const double* z, const PcvRouted& routed, const uint8_t* color,
                            uint32_t color_stride, const float* intensity, uint32_t* rank, void* payload,
                            uint32_t* inten_bits, uint8_t* depth_grid /* pcv_spec_depth_grid_bytes() of scratch, or null */,
                            void* wide, uint32_t* pool_ctr /* kPcvPoolRegions zeroed counters: entries of `wide` handed out per region */,
                            const uint32_t* tree_info /* device: [0] = number of T'' nodes (spec_tree_scan_kernel's info block) */,
                            bool color_late, uint32_t* zero, size_t zero_words) {
  if (n == 0) return;
  PcvProf prof(ctx, PCV_K_SPEC_ENCODE);
  (void)tree_info;
  constexpr int kBlock = 512;  // tiles of 1 024 points: one pool region, 26 KB of LDS, four workgroups per CU
  const dim3 grid((unsigned)((n + 2 * kBlock - 1) / (2 * kBlock)));
  const float cells = lv.edge[0] > 0.0 ? (float)((double)(1 << kGridBits) / lv.edge[0]) : 0.f;
  const uint32_t pool_cap = (uint32_t)pcv_pool_region_entries(n);
  if (depth_grid)  // (zero: 16-byte aligned, a multiple of four words — or the caller clears it itself)
    hipLaunchKernelGGL(spec_depth_grid_kernel, dim3((1u << (3 * kGridBits)) / 256), dim3(256), 0, ctx->stream, walk, depth_grid, (uint4*)zero,
                       (uint32_t)(zero_words / 4));
  // the first kTop walk records (levels 0-3 of T'' and the start of level 4: the table is level-major) are mirrored in LDS: with
  // the Float32 code steps a shallow level step is shorter than the L2 round trip of its child gather. One call, 100 M points:
  // 0 / 512 / 1 024 / 1 536 / 2 048 / 3 072 records -> 1.92-1.95 / 1.88 / 1.88 / 1.85 / 1.83-1.85 / 2.11 ms (3 072: 39 KB of LDS,
  // registers); profiles/r05_ab_chain_pass_walk_top_in_lds.json. PCV_CHAIN_TOP (libpcv_hip_exp.so): another size, 0 = none.
  constexpr int kTop = 2048;
#define PCV_CHAIN_TOP_LAUNCH(RAWIN, T)                                                                                                  \
  hipLaunchKernelGGL((chain_pass_kernel<true, RAWIN, kBlock, T>), grid, dim3(kBlock), 0, ctx->stream, lv, walk, n, x, y, z, routed, color, \
                     color_stride, intensity, rank, (uint4*)payload, inten_bits, depth_grid, cells, (uint4*)wide, pool_ctr, pool_cap)
#ifdef PCV_EXPERIMENTS
  static const int chain_top = [] {
    const char* e = pcv_experiment("PCV_CHAIN_TOP");
    return e ? atoi(e) : kTop;
  }();
  if (!routed.oct && chain_top != kTop) {
    switch (chain_top) {
      case 0: PCV_CHAIN_TOP_LAUNCH(true, 0); return;
      case 512: PCV_CHAIN_TOP_LAUNCH(true, 512); return;
      case 1024: PCV_CHAIN_TOP_LAUNCH(true, 1024); return;
      case 1536: PCV_CHAIN_TOP_LAUNCH(true, 1536); return;
      case 3072: PCV_CHAIN_TOP_LAUNCH(true, 3072); return;
      default: break;
    }
  }
#endif
#ifdef PCV_EXPERIMENTS  // (PCV_COLOR_LATE=1: measured, slower overall, not instantiated in libpcv_hip.so)
#define PCV_CHAIN_LATE_LAUNCH(RAWIN)                                                                                                     \
  hipLaunchKernelGGL((chain_pass_kernel<true, RAWIN, kBlock, kTop, 1>), grid, dim3(kBlock), 0, ctx->stream, lv, walk, n, x, y, z, routed, color, \
                     color_stride, intensity, rank, (uint4*)payload, inten_bits, depth_grid, cells, (uint4*)wide, pool_ctr, pool_cap)
  if (color_late && wide) {  // (12-byte records only: the 20-byte form fetches the colour in its record epilogue)
    if (!routed.oct) PCV_CHAIN_LATE_LAUNCH(true);
    else PCV_CHAIN_LATE_LAUNCH(false);
    return;
  }
#undef PCV_CHAIN_LATE_LAUNCH
#endif
  (void)color_late;
  if (!routed.oct) PCV_CHAIN_TOP_LAUNCH(true, kTop);
  else PCV_CHAIN_TOP_LAUNCH(false, kTop);
#undef PCV_CHAIN_TOP_LAUNCH
}

size_t pcv_spec_depth_grid_bytes() { return (size_t)1 << (3 * kGridBits); }

__global__ __launch_bounds__(256) void join_color_kernel(uint64_t n, const uint8_t* __restrict__ color, uint32_t color_stride,
                                                          uint32_t* __restrict__ keys, uint2* __restrict__ payload) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t rgb = pcv_load_rgb(color + i * color_stride, i + 1 < n);
  keys[i] |= rgb >> 16;
  payload[i].y |= (rgb & 0xffffu) << 16;
}
void pcv_launch_join_color(pcv_ctx* ctx, uint64_t n, const uint8_t* color, uint32_t color_stride, uint32_t* keys, void* payload_uint2) {
  if (n == 0) return;
  hipLaunchKernelGGL(join_color_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, color, color_stride, keys,
                     (uint2*)payload_uint2);
}

void pcv_launch_rank_hist(pcv_ctx* ctx, const uint32_t* rank, uint64_t n, uint32_t num_bins, uint32_t* counts, int shift) {
  if (n == 0 || num_bins == 0) return;
  // one workgroup per CU-ish: 512 workgroups of 1024 lanes; the chunk is a multiple of 4096 keys
  uint64_t chunk = (n + 511) / 512;
  chunk = (chunk + 4095) & ~(uint64_t)4095;
  const unsigned groups = (unsigned)((n + chunk - 1) / chunk);
  for (uint32_t base = 0; base < num_bins; base += kHistBins) {
    const uint32_t nb = num_bins - base < (uint32_t)kHistBins ? num_bins - base : (uint32_t)kHistBins;
    PcvProf prof(ctx, PCV_K_RANK_HIST);
    hipLaunchKernelGGL(rank_hist_kernel, dim3(groups), dim3(1024), (size_t)nb * 4, ctx->stream, rank, n, chunk, base, nb, counts, shift,
                       (uint32_t*)nullptr, 0u);
  }
}

// The same count over the workgroups and chunks of the record sort (pcv_sort_rec12_geometry), every workgroup's histogram kept
// as a row: the sort's first pass derives its digit histogram from the rows and the rank map (pcv_sort.hip) instead of
// reading the keys once more. Up to 32 768 bins per launch (128 KB of LDS, one workgroup per CU: the sort's own occupancy);
// bigger trees (1 B points: ~50 000 predicted nodes) take one launch per 32 768 bins. num_bins <= pcv_rank_hist_max_bins().
uint32_t pcv_rank_hist_max_bins() { return 1u << 18; }
void pcv_launch_rank_hist_rows(pcv_ctx* ctx, const uint32_t* rank, uint64_t n, uint32_t num_bins, uint32_t* counts, int shift,
                               int groups, uint64_t chunk, uint32_t* rows) {
  if (n == 0 || num_bins == 0) return;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kHistBinsRows * 4) == hipSuccess;
  (void)ok;
  for (uint32_t base = 0; base < num_bins; base += kHistBinsRows) {
    const uint32_t nb = num_bins - base < (uint32_t)kHistBinsRows ? num_bins - base : (uint32_t)kHistBinsRows;
    PcvProf prof(ctx, PCV_K_RANK_HIST);
    hipLaunchKernelGGL(rank_hist_kernel, dim3(groups), dim3(1024), (size_t)nb * 4, ctx->stream, rank, n, chunk, base, nb, counts, shift, rows,
                       num_bins);
  }
}

size_t pcv_cont_range_bytes() { return sizeof(PcvContRange); }
void pcv_fill_cont_range(void* dst, uint32_t from_level, uint32_t to_level, const double mn[3]) {
  PcvContRange* r = reinterpret_cast<PcvContRange*>(dst);
  *r = PcvContRange{from_level, to_level, 0u, 0u, {mn[0], mn[1], mn[2]}, 0.0};
}
void pcv_launch_spec_continue(pcv_ctx* ctx, const PcvLevels& lv, const void* ranges, const PcvSettleItem* items, uint32_t num_items,
                              void* sorted_payload, void* wide) {
  if (num_items == 0) return;
  PcvProf prof(ctx, PCV_K_SPEC_CONTINUE);
  if (wide)
    hipLaunchKernelGGL((spec_continue_kernel<true>), dim3(num_items), dim3(256), 0, ctx->stream, lv, (const PcvContRange*)ranges, items,
                       (uint4*)sorted_payload, (uint4*)wide);
  else
    hipLaunchKernelGGL((spec_continue_kernel<false>), dim3(num_items), dim3(256), 0, ctx->stream, lv, (const PcvContRange*)ranges, items,
                       (uint4*)sorted_payload, (uint4*)nullptr);
}

void pcv_launch_spec_replay(pcv_ctx* ctx, const PcvLevels& lv, const void* ranges, uint32_t num_ranges, uint32_t total,
                            const double* x, const double* y, const double* z, const PcvRouted& routed, void* sorted_payload,
                            void* wide, uint32_t pool_cap) {
  if (total == 0 || num_ranges == 0) return;
  PcvProf prof(ctx, PCV_K_SPEC_REPLAY);
  const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)total + 255) / 256, 8192);
  hipLaunchKernelGGL(spec_replay_kernel, dim3(blocks), dim3(256), 0, ctx->stream, lv, (const PcvFixRange*)ranges, num_ranges, total,
                     x, y, z, routed, (uint4*)sorted_payload, (uint4*)wide, pool_cap);
}

size_t pcv_climber_bytes(uint64_t num_climbers) { return (size_t)(num_climbers + 1) * sizeof(PcvClimber); }
bool pcv_climb16_enabled() {
  static const bool climb16_on = [] {  // PCV_CLIMB16=0 (libpcv_hip_exp.so): 32-byte climber records everywhere
    const char* e = pcv_experiment("PCV_CLIMB16");
    return !e || atoi(e) != 0;
  }();
  return climb16_on;
}

void pcv_launch_promote_encode(pcv_ctx* ctx, const PcvLevels& lv, const PcvPromoteTables& pt, uint64_t n,
                               const uint32_t* rank, const void* payload, const uint32_t* cx_hi, const uint32_t* cy_hi,
                               const uint32_t* cz_hi, const uint32_t* inten_bits, const uint32_t* climb_base,
                               uint32_t num_climbers, void* climbers, uint8_t* xyz_blob, uint8_t* rgb_blob,
                               uint8_t* inten_blob, const void* wide, const PcvSettleItem* items, uint32_t num_items,
                               const PcvSettleItem* climb_items, uint32_t num_climb_items, const void* cont_ranges) {
  if (n == 0) return;
  PromoteOut o{xyz_blob, rgb_blob, inten_blob};
  // 16-byte climbers: leaf-wise kernels, no intensity plane, no Float64 high words
#ifdef PCV_EXPERIMENTS
  static const bool wide_mask_set = [] {
    if (const char* e = pcv_experiment("PCV_WIDE_MASK")) {
      const uint32_t m = (uint32_t)strtoul(e, nullptr, 0);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(pcv_exp_wide_mask), &m, sizeof(m));
    }
    return true;
  }();
  (void)wide_mask_set;
#endif
  const bool climb16 = pcv_climb16_enabled() && items && climb_items && !inten_bits && !cx_hi;
  if (items) {
    uint32_t settle_grid = num_items;
#ifdef PCV_EXPERIMENTS
    static const bool xcd = [] {
      const char* e = pcv_experiment("PCV_SETTLE_XCD");
      return e && atoi(e) != 0;
    }();
    if (xcd) {
      settle_grid = (num_items + 7u) / 8u * 8u;
      (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(pcv_exp_settle_items), &num_items, sizeof(num_items), 0, hipMemcpyHostToDevice, ctx->stream);
    }
#endif
    PcvProf prof(ctx, PCV_K_PROMOTE_ENCODE);
#define PCV_SETTLE_LEAF(C, K, W)                                                                                              \
  hipLaunchKernelGGL((promote_settle_leaf_kernel<C, K>), dim3(settle_grid), dim3(256), 0, ctx->stream, pt, items, rank,         \
                     (const uint4*)payload, cx_hi, cy_hi, cz_hi, inten_bits, climb_base, (PcvClimber*)climbers, o, (const uint4*)(W), lv, \
                     (const PcvContRange*)cont_ranges)
    if (num_items && wide && climb16) PCV_SETTLE_LEAF(true, true, wide);
    else if (num_items && wide) PCV_SETTLE_LEAF(true, false, wide);
    else if (num_items && climb16) PCV_SETTLE_LEAF(false, true, nullptr);
    else if (num_items) PCV_SETTLE_LEAF(false, false, nullptr);
#undef PCV_SETTLE_LEAF
  } else {
    PcvProf prof(ctx, PCV_K_PROMOTE_ENCODE);
    static const int slots = [] {  // PCV_SETTLE_SLOTS (experiments): 1, 2 or 4 sorted slots per lane
      const char* e = pcv_experiment("PCV_SETTLE_SLOTS");
      return e ? atoi(e) : 2;
    }();
#define PCV_SETTLE(S, C)                                                                                                 \
  hipLaunchKernelGGL((promote_settle_kernel<S, C>), dim3((unsigned)((n + 256 * S - 1) / (256 * S))), dim3(256), 0, ctx->stream, pt, \
                     n, rank, (const uint4*)payload, cx_hi, cy_hi, cz_hi, inten_bits, climb_base, (PcvClimber*)climbers, o,       \
                     (const uint4*)wide)
    if (wide && slots == 1) PCV_SETTLE(1, true);
    else if (wide && slots == 4) PCV_SETTLE(4, true);
    else if (wide) PCV_SETTLE(2, true);
    else if (slots == 1) PCV_SETTLE(1, false);
    else if (slots == 4) PCV_SETTLE(4, false);
    else PCV_SETTLE(2, false);
#undef PCV_SETTLE
  }
  if (num_climbers && climb_items) {
    PcvProf prof(ctx, PCV_K_PROMOTE_CLIMB);
    if (num_climb_items && climb16)
      hipLaunchKernelGGL(promote_climb_leaf_kernel<true>, dim3(num_climb_items), dim3(256), 0, ctx->stream, pt, climb_items,
                         (const PcvClimber*)climbers, cx_hi, cy_hi, cz_hi, o);
    else if (num_climb_items)
      hipLaunchKernelGGL(promote_climb_leaf_kernel<false>, dim3(num_climb_items), dim3(256), 0, ctx->stream, pt, climb_items,
                         (const PcvClimber*)climbers, cx_hi, cy_hi, cz_hi, o);
  } else if (num_climbers) {
    PcvProf prof(ctx, PCV_K_PROMOTE_CLIMB);
    hipLaunchKernelGGL(promote_climb_kernel, dim3((num_climbers + 255) / 256), dim3(256), 0, ctx->stream, pt, num_climbers,
                       (const PcvClimber*)climbers, cx_hi, cy_hi, cz_hi, o);
  }
}
