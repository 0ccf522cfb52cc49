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
#include "pcv_chain_dev.h"

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

struct PromoteOut {
  uint8_t* xyz_blob;
  uint8_t* rgb_blob;
  uint8_t* inten_blob;
};

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
  uint32_t slot = j;
  const uint32_t enc = cur.enc;
  if (cur.parent != 0xffffffffu) {
    slot = j - (j >> 3) - 1u;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      code[a] = pcv_encode_coord(enc, pcv_decode_coord(enc, code[a], cur.mn[a], cur.edge), cur.mn[a], cur.edge, PcvRecip{cur.inv_edge, cur.inv_edge_lo});
  }
  uint8_t* dst = o.xyz_blob + cur.xyz_off;
  switch (enc) {
    case PCV_ENC_UINT8: {
      uint8_t* d = dst + (uint64_t)slot * 3;
      d[0] = (uint8_t)code[0];
      d[1] = (uint8_t)code[1];
      d[2] = (uint8_t)code[2];
      break;
    }
    case PCV_ENC_UINT16: {
      uint16_t* d = reinterpret_cast<uint16_t*>(dst) + (uint64_t)slot * 3;
      d[0] = (uint16_t)code[0];
      d[1] = (uint16_t)code[1];
      d[2] = (uint16_t)code[2];
      break;
    }
    case PCV_ENC_FLOAT32: {
      uint32_t* d = reinterpret_cast<uint32_t*>(dst) + (uint64_t)slot * 3;
      d[0] = (uint32_t)code[0];
      d[1] = (uint32_t)code[1];
      d[2] = (uint32_t)code[2];
      break;
    }
    default: {
      uint64_t* d = reinterpret_cast<uint64_t*>(dst) + (uint64_t)slot * 3;
      d[0] = code[0];
      d[1] = code[1];
      d[2] = code[2];
      break;
    }
  }
  const uint64_t pidx = cur.point_off + slot;
  const uint32_t c = pay.w;
  uint8_t* cd = o.rgb_blob + pidx * 3;
  cd[0] = (uint8_t)c;
  cd[1] = (uint8_t)(c >> 8);
  cd[2] = (uint8_t)(c >> 16);
  if (o.inten_blob) reinterpret_cast<uint32_t*>(o.inten_blob)[pidx] = inten;
}

// K6 runs as two kernels over the sorted records. Seven of eight points stay in their leaf: `settle` streams over all
// slots (two per lane, both record chains started before either is consumed) and finishes those with straight-line
// code. The every-8th points climb a data-dependent number of levels (decode + encode per level): `climb` visits one
// octet of slots per lane, so its waves are full of climbers instead of carrying seven idle lanes through the loop.
__global__ __launch_bounds__(256) void promote_settle_kernel(
    PcvPromoteTables pt, uint64_t n, const uint32_t* __restrict__ rank, const uint4* __restrict__ payload,
    const uint32_t* __restrict__ cx_hi, const uint32_t* __restrict__ cy_hi, const uint32_t* __restrict__ cz_hi,
    const uint32_t* __restrict__ inten_bits, PromoteOut o) {
  const uint64_t s0 = (uint64_t)blockIdx.x * 512 + threadIdx.x;
  const uint64_t s1 = s0 + 256;
  if (s0 >= n) return;
  const bool two = s1 < n;
  const uint32_t r0 = rank[s0];
  const uint32_t r1 = two ? rank[s1] : 0u;
  const uint4 p0 = payload[s0];
  const uint4 p1 = two ? payload[s1] : make_uint4(0, 0, 0, 0);
  uint32_t h0[3] = {0, 0, 0}, h1[3] = {0, 0, 0}, i0 = 0, i1 = 0;
  if (cx_hi) {
    h0[0] = cx_hi[s0];
    h0[1] = cy_hi[s0];
    h0[2] = cz_hi[s0];
    if (two) {
      h1[0] = cx_hi[s1];
      h1[1] = cy_hi[s1];
      h1[2] = cz_hi[s1];
    }
  }
  if (inten_bits) {
    i0 = inten_bits[s0];
    if (two) i1 = inten_bits[s1];
  }
  const PcvNodeRec c0 = pt.leaf_rec[r0];
  const PcvNodeRec c1 = pt.leaf_rec[r1];
  const bool stay0 = c0.parent == 0xffffffffu || (((uint32_t)s0 - c0.lo) & 7u) != 0;
  const bool stay1 = two && (c1.parent == 0xffffffffu || (((uint32_t)s1 - c1.lo) & 7u) != 0);
  if (stay0) promote_one<false>(pt, s0, c0, p0, h0[0], h0[1], h0[2], i0, o);
  if (stay1) promote_one<false>(pt, s1, c1, p1, h1[0], h1[1], h1[2], i1, o);
}

__device__ __forceinline__ void climb_slot(const PcvPromoteTables& pt, uint64_t s, const PcvNodeRec& rec,
                                           const uint4* __restrict__ payload, const uint32_t* __restrict__ cx_hi,
                                           const uint32_t* __restrict__ cy_hi, const uint32_t* __restrict__ cz_hi,
                                           const uint32_t* __restrict__ inten_bits, const PromoteOut& o) {
  const uint4 p = payload[s];
  uint32_t h[3] = {0, 0, 0}, in = 0;
  if (cx_hi) {
    h[0] = cx_hi[s];
    h[1] = cy_hi[s];
    h[2] = cz_hi[s];
  }
  if (inten_bits) in = inten_bits[s];
  promote_one<true>(pt, s, rec, p, h[0], h[1], h[2], in, o);
}

__global__ __launch_bounds__(256) void promote_climb_kernel(
    PcvPromoteTables pt, uint64_t n, const uint32_t* __restrict__ rank, const uint4* __restrict__ payload,
    const uint32_t* __restrict__ cx_hi, const uint32_t* __restrict__ cy_hi, const uint32_t* __restrict__ cz_hi,
    const uint32_t* __restrict__ inten_bits, PromoteOut o) {
  const uint64_t s0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;  // this lane's octet of sorted slots
  if (s0 >= n) return;
  uint32_t r[8];
  if (s0 + 8 <= n) {  // rank comes from the pool (256-byte aligned) and s0 is a multiple of 8: two 16-byte loads
    const uint4 a = *reinterpret_cast<const uint4*>(rank + s0);
    const uint4 b = *reinterpret_cast<const uint4*>(rank + s0 + 4);
    r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = s0 + k < n ? rank[s0 + k] : 0xffffffffu;
  }
  const PcvNodeRec first = pt.leaf_rec[r[0]];
  if (r[7] == r[0]) {  // ranks are sorted: the whole octet lies in one leaf, which holds exactly one climber of it
    if (first.parent != 0xffffffffu)
      climb_slot(pt, s0 + ((first.lo - (uint32_t)s0) & 7u), first, payload, cx_hi, cy_hi, cz_hi, inten_bits, o);
    return;
  }
  // a leaf ends inside the octet (or the input ends): look at every slot
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    if (r[k] == 0xffffffffu) break;
    const PcvNodeRec rec = r[k] == r[0] ? first : pt.leaf_rec[r[k]];
    const uint64_t s = s0 + k;
    if (rec.parent != 0xffffffffu && (((uint32_t)s - rec.lo) & 7u) == 0)
      climb_slot(pt, s, rec, payload, cx_hi, cy_hi, cz_hi, inten_bits, o);
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

void pcv_launch_promote_encode(pcv_ctx* ctx, const PcvLevels& lv, const PcvPromoteTables& pt, uint64_t n,
                               const uint32_t* rank, const void* payload, const uint32_t* cx_hi, const uint32_t* cy_hi,
                               const uint32_t* cz_hi, const uint32_t* inten_bits, uint8_t* xyz_blob, uint8_t* rgb_blob,
                               uint8_t* inten_blob) {
  if (n == 0) return;
  PromoteOut o{xyz_blob, rgb_blob, inten_blob};
  {
    PcvProf prof(ctx, PCV_K_PROMOTE_ENCODE);
    hipLaunchKernelGGL(promote_settle_kernel, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, ctx->stream, pt, n, rank,
                       (const uint4*)payload, cx_hi, cy_hi, cz_hi, inten_bits, o);
  }
  {
    PcvProf prof(ctx, PCV_K_PROMOTE_CLIMB);
    hipLaunchKernelGGL(promote_climb_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, ctx->stream, pt, n, rank,
                       (const uint4*)payload, cx_hi, cy_hi, cz_hi, inten_bits, o);
  }
}
