// pcv_settle_dev.h — device code shared by the kernels that finish sorted records: the final rewrite and store of a point that
// stays in its node (K6, generation.rs:195-253,335-387). Used by `settle` / `climb` (pcv_encode.hip) and by the
// record sort's last pass when it settles the leaves' points itself (pcv_sort.hip). Include after pcv_chain_dev.h.
#pragma once
#include "pcv_chain_dev.h"

#ifndef PCV_SETTLE_DIAG
#define PCV_SETTLE_DIAG 0
#endif

namespace {

struct PromoteOut {
  uint8_t* xyz_blob;
  uint8_t* rgb_blob;
  uint8_t* inten_blob;
};

// The point at position j of node `cur`'s stream stays there: final rewrite (encode(decode(code)) at the node's own level —
// not idempotent, SURVEY F5 — unless the node is the root, which keeps what it receives) and the stores, as straight-line
// code for one encoding.
template <int ENC, bool CLIMB>
__device__ __forceinline__ void promote_final(const PcvNodeRec& cur, uint32_t j, uint64_t (&code)[3], uint32_t rgb,
                                              uint32_t inten, const PromoteOut& o) {
  uint32_t slot = j;
  if (cur.parent != 0xffffffffu) {
    slot = j - (j >> 3) - 1u;
#if PCV_SETTLE_DIAG != 2  // (timing experiments, tools/build_variants.sh: 2 = no re-encode, 1 = no stores; never shipped)
    if ((ENC == PCV_ENC_UINT8 || ENC == PCV_ENC_UINT16) && cur.inv_edge != 0.0) {
      // integer codes in a tame cube (the host zeroes inv_edge otherwise): the decoded position lies inside the cube, so
      // the exact constant-divisor division needs no range check (pcv_div_const<false>)
#pragma unroll
      for (int a = 0; a < 3; ++a)
        code[a] = pcv_fix_encode<false>(pcv_decode_coord(ENC, code[a], cur.mn[a], cur.edge), cur.mn[a], cur.edge,
                                        PcvRecip{cur.inv_edge, cur.inv_edge_lo}, ENC == PCV_ENC_UINT8 ? 255.0 : 65535.0);
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a)
        code[a] = pcv_encode_coord(ENC, pcv_decode_coord(ENC, code[a], cur.mn[a], cur.edge), cur.mn[a], cur.edge,
                                   PcvRecip{cur.inv_edge, cur.inv_edge_lo});
    }
#endif
  }
#if PCV_SETTLE_DIAG == 1
  if (!CLIMB && code[0] != 0x7fffffffffffull) return;
#endif
  uint8_t* dst = o.xyz_blob + cur.xyz_off;
  if (ENC == PCV_ENC_UINT8) {
    uint8_t* d = dst + (uint64_t)slot * 3;
    d[0] = (uint8_t)code[0];
    d[1] = (uint8_t)code[1];
    d[2] = (uint8_t)code[2];
  } else if (ENC == PCV_ENC_UINT16) {
    uint16_t* d = reinterpret_cast<uint16_t*>(dst) + (uint64_t)slot * 3;
    d[0] = (uint16_t)code[0];
    d[1] = (uint16_t)code[1];
    d[2] = (uint16_t)code[2];
  } else if (ENC == PCV_ENC_FLOAT32) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst) + (uint64_t)slot * 3;
    d[0] = (uint32_t)code[0];
    d[1] = (uint32_t)code[1];
    d[2] = (uint32_t)code[2];
  } else {
    uint64_t* d = reinterpret_cast<uint64_t*>(dst) + (uint64_t)slot * 3;
    d[0] = code[0];
    d[1] = code[1];
    d[2] = code[2];
  }
  const uint64_t pidx = cur.point_off + slot;
  uint8_t* cd = o.rgb_blob + pidx * 3;
  cd[0] = (uint8_t)rgb;
  cd[1] = (uint8_t)(rgb >> 8);
  cd[2] = (uint8_t)(rgb >> 16);
  if (o.inten_blob) reinterpret_cast<uint32_t*>(o.inten_blob)[pidx] = inten;
}

// One sorted slot: climb, final encode, store. CLIMB = false: the caller knows the point stays in its leaf.
template <bool CLIMB>
__device__ __forceinline__ void promote_one(const PcvPromoteTables& pt, uint64_t s, PcvNodeRec cur, uint4 pay,
                                            uint32_t hx, uint32_t hy, uint32_t hz, uint32_t inten, const PromoteOut& o) {
  uint32_t j = (uint32_t)s - cur.lo;
  uint64_t code[3] = {pay.x | ((uint64_t)hx << 32), pay.y | ((uint64_t)hy << 32), pay.z | ((uint64_t)hz << 32)};
  // climb while this point is an every-8th element of its node's stream
  while (CLIMB && cur.parent != 0xffffffffu && (j & 7u) == 0) {
    const PcvNodeRec par = pt.node_rec[cur.parent];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double q = pcv_decode_coord(cur.enc, code[a], cur.mn[a], cur.edge);
      code[a] = pcv_encode_coord(par.enc, q, par.mn[a], par.edge, PcvRecip{par.inv_edge, par.inv_edge_lo});
    }
    j = cur.child_off + (j >> 3);
    cur = par;
  }
  switch (cur.enc) {  // the node's encoding is wave-uniform in `settle` (one leaf per workgroup): one scalar branch
    case PCV_ENC_UINT8: return promote_final<PCV_ENC_UINT8, CLIMB>(cur, j, code, pay.w, inten, o);
    case PCV_ENC_UINT16: return promote_final<PCV_ENC_UINT16, CLIMB>(cur, j, code, pay.w, inten, o);
    case PCV_ENC_FLOAT32: return promote_final<PCV_ENC_FLOAT32, CLIMB>(cur, j, code, pay.w, inten, o);
    default: return promote_final<PCV_ENC_FLOAT64, CLIMB>(cur, j, code, pay.w, inten, o);
  }
}

// a climber as `settle` (or the record sort's settling pass) hands it to `climb` when the 16-byte payload alone will not do
// (an intensity plane, Float64 high words): codes + colour, its leaf, its sorted slot, its intensity
struct alignas(16) PcvClimber {
  uint4 pay;
  uint32_t rank, slot, inten, pad;
};
}  // namespace
