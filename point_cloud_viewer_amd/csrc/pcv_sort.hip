// pcv_sort.hip — K3: stable LSD radix sort for gfx950 (wave64), 8-bit digits, LDS histograms.
//
// The reference never sorts: it partitions every node's stream into 8 child files level by level
// (src/octree/generation.rs:58-126: 8 clones + 8 `retain`s per batch). On the GPU the same *stable* grouping
// is one radix sort of the path keys (and later of leaf-rank records), SURVEY.md §8a R7 / F11.
//
// Structure per 8-bit pass (reduce-then-scan, no inter-workgroup spinning):
//   upsweep   : G workgroups, each counts the digits of its contiguous chunk in per-wave LDS histograms
//   scan      : one workgroup per digit scans that digit's G counts; digit totals are scanned in the downsweep
//   downsweep : the same G workgroups walk their chunk tile by tile; inside a tile each wave ranks its keys
//               with ballot-built peer masks (64-lane match-any), a 256-entry LDS scan orders the digits,
//               keys (and the record payload) are staged through LDS so global stores go out as runs.
// Two downsweep kernels: keys only (u32/u64, 16 keys per lane, next tile prefetched into registers) and records
// (u32 key + one 16-byte payload word per key, 8 per lane, + optional extra 4-byte planes).
// HBM traffic per pass and key: sizeof(key) (upsweep) + 2 * sizeof(key) + 2 * payload bytes.
#include <cstdlib>

#include "pcv_internal.h"
#include "pcv_settle_dev.h"

#ifndef PCV_FUSE_DIAG
#define PCV_FUSE_DIAG 0
#endif
#define PCV_SPEC_INDEX_MASK_SORT 0x3fffffffu  // == PCV_SPEC_INDEX_MASK (pcv_spec.h)

namespace {

constexpr int kBlock = 256;  // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kRadix = 256;
#ifndef PCV_SORT_GROUPS
#define PCV_SORT_GROUPS 1024
#endif
// records per lane and tile of the record kernel: 16 = tiles of 4 096 records (86 KB of LDS, one workgroup per CU) beat 8
// (three workgroups per CU) by 0.1-0.15 ms per pass at 100 M records — twice as long write runs per digit, a third of
// the concurrent write streams
#ifndef PCV_KPT_REC
#define PCV_KPT_REC 16
#endif
#ifndef PCV_KEYS_WAVES
#define PCV_KEYS_WAVES 4
#endif
constexpr int kMaxGroups = PCV_SORT_GROUPS;
#ifndef PCV_KEYS_KPT
#define PCV_KEYS_KPT 16
#endif
constexpr int kKptKeys = PCV_KEYS_KPT;  // keys-only kernel: keys per lane per tile
constexpr int kKptRec = PCV_KPT_REC;    // record kernel
constexpr int lcm_kpt(int a, int b) {
  int x = a, y = b;
  while (y) {
    const int t = x % y;
    x = y;
    y = t;
  }
  return a / x * b;
}
constexpr int kTileUnit = kBlock * lcm_kpt(kKptKeys, kKptRec);  // chunk granularity (multiple of both tile sizes)

struct SortGeom {
  uint64_t n;
  uint64_t chunk;  // keys per workgroup, multiple of kTileUnit
  int groups;
};

SortGeom make_geom(uint64_t n, uint64_t unit = kTileUnit) {
  SortGeom g;
  g.n = n;
  uint64_t tiles = (n + unit - 1) / unit;
  uint64_t tiles_per_group = (tiles + kMaxGroups - 1) / kMaxGroups;
  if (tiles_per_group == 0) tiles_per_group = 1;
  g.chunk = tiles_per_group * unit;
  g.groups = (int)((n + g.chunk - 1) / g.chunk);
  if (g.groups < 1) g.groups = 1;
  return g;
}

// One histogram update per group of lanes holding the same digit: the lanes are matched with ballots (as in the
// downsweep ranking) and only the first of each group issues the LDS add, with the group size. Plain per-lane LDS
// atomics serialise on equal addresses, and the digits of path keys / leaf ranks are heavily skewed (the upper
// digits take a few dozen values), which cost up to 60 % over uniform keys.
__device__ __forceinline__ void count_digit(uint32_t* __restrict__ wh, uint32_t d, uint64_t valid_mask, bool valid) {
  uint32_t plo = (uint32_t)valid_mask, phi = (uint32_t)(valid_mask >> 32);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(d), "n"(b));
    const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
    plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, (uint32_t)m, 0x90);  // p & ~(ballot ^ m)
    phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
  }
  const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
  if (valid && below == 0) atomicAdd(&wh[d], (uint32_t)(__popc(plo) + __popc(phi)));
}

// kPlain: one LDS add per key instead of the ballot match — for digits that are spread evenly over the wave (the upper
// digit of the record sort after the pass on the lower one: 0.132 -> 0.115 ms at 100 M records; the match wins on the
// skewed digits of the first pass and of the path keys)
template <typename KeyT, bool kPlain = false>
__global__ __launch_bounds__(kBlock) void upsweep_kernel(const KeyT* __restrict__ keys, uint64_t n, uint64_t chunk,
                                                          int groups, int shift, uint32_t mask,
                                                          uint32_t* __restrict__ hist /* [256][groups] */) {
  __shared__ uint32_t wh[kWaves][kRadix];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kWaves * kRadix; i += kBlock) (&wh[0][0])[i] = 0;
  __syncthreads();
  const uint64_t begin = (uint64_t)blockIdx.x * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;
  constexpr int kVec = 16 / sizeof(KeyT);  // keys per 16-byte load
  typedef KeyT VecT __attribute__((ext_vector_type(kVec)));
  // chunk is a multiple of the tile and buffers come from the pool (256-B aligned) => 16-byte loads are aligned
  uint64_t i = begin + (uint64_t)threadIdx.x * kVec;
  constexpr uint64_t kStep = (uint64_t)kBlock * kVec;
  // four 16-byte loads in flight per lane: the loop is latency bound otherwise
  for (; i + 3 * kStep + kVec <= end; i += 4 * kStep) {
    VecT v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const VecT*>(keys + i + u * kStep);
    // the loop condition is not wave-uniform in the last iterations of a chunk: match only the lanes that are here
    const uint64_t here = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < kVec; ++k) {
        if (kPlain) atomicAdd(&wh[wave][(uint32_t)(v[u][k] >> shift) & mask], 1u);
        else count_digit(wh[wave], (uint32_t)(v[u][k] >> shift) & mask, here, true);
      }
  }
  for (; i + kVec <= end; i += kStep) {
    VecT v = *reinterpret_cast<const VecT*>(keys + i);
    const uint64_t here = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
    for (int k = 0; k < kVec; ++k) {
      if (kPlain) atomicAdd(&wh[wave][(uint32_t)(v[k] >> shift) & mask], 1u);
      else count_digit(wh[wave], (uint32_t)(v[k] >> shift) & mask, here, true);
    }
  }
  for (; i < end; ++i) atomicAdd(&wh[wave][(uint32_t)(keys[i] >> shift) & mask], 1u);  // ragged tail (< kVec keys)
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += kBlock) {
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += wh[w][d];
    hist[(uint64_t)d * groups + blockIdx.x] = s;
  }
}

// First record pass of the single-chain build: the upsweep reads every rank anyway, so it also translates it on the way
// — rank := map[rank], written back in place — and counts the digits of the MAPPED ranks: one pass over the ranks
// instead of two; the downsweep is the ordinary one. The payloads are not touched: a record already carries the codes
// its true leaf needs (pcv_spec.h), except for the rare leaves the map flags PCV_SPEC_MAP_REPLAY (bit 30), whose points
// leave their input index in the first payload word for the replay after the sort.
// kMapLds: the map (one entry per predicted leaf; 30 KB for a 100 M-point tree) is copied into LDS first — eight
// dependent lookups per lane and iteration then cost LDS latency instead of a trip to the vector L1 / L2 that the
// streaming keys keep evicting it from. Static + dynamic LDS stay inside the 64 KB a kernel gets without opting in.
constexpr uint32_t kMapLdsEntries = 15000;  // 60 000 bytes next to the 4 KB of counters
// kCompact (12-byte records, pcv_internal.h): key = rank << 8 | blue, payload = uint2; `shift` is the digit's position
// inside the KEY (8 + its position inside the rank).
template <bool kMapLds, bool kCompact>
__global__ __launch_bounds__(kBlock) void upsweep_map_kernel(uint32_t* __restrict__ keys, uint64_t n, uint64_t chunk, int groups,
                                                              int shift, uint32_t mask, uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ gmap, uint32_t map_entries,
                                                              void* __restrict__ payload_v) {
  __shared__ uint32_t wh[kWaves][kRadix];
  extern __shared__ uint32_t smap[];  // kMapLds: map_entries words (dynamic, so small maps keep the occupancy)
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kWaves * kRadix; i += kBlock) (&wh[0][0])[i] = 0;
  if (kMapLds)
    for (uint32_t i = threadIdx.x; i < map_entries; i += kBlock) smap[i] = gmap[i];
  const uint32_t* map = kMapLds ? smap : gmap;
  __syncthreads();
  const uint64_t begin = (uint64_t)blockIdx.x * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;
  auto one = [&](uint64_t idx, uint32_t old, uint32_t m) -> uint32_t {
    if (__builtin_expect((m & (1u << 30)) != 0u, 0)) {  // replay: the first payload word becomes the input index
      if (kCompact) reinterpret_cast<uint32_t*>(reinterpret_cast<uint2*>(payload_v) + idx)[0] = (uint32_t)idx;
      else reinterpret_cast<uint32_t*>(reinterpret_cast<uint4*>(payload_v) + idx)[0] = (uint32_t)idx;
    }
    return kCompact ? (((m & PCV_SPEC_INDEX_MASK_SORT) << 8) | (old & 0xffu)) : (m & PCV_SPEC_INDEX_MASK_SORT);
  };
  constexpr int kKeyShift = kCompact ? 8 : 0;  // position of the predicted-leaf rank inside the key
  uint64_t i = begin + (uint64_t)threadIdx.x * 4;
  constexpr uint64_t kStep = (uint64_t)kBlock * 4;
  for (; i + 3 * kStep + 4 <= end; i += 4 * kStep) {  // four 16-byte loads and their sixteen map lookups in flight per lane
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(keys + i + u * kStep);
    uint32_t r[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t o[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      uint32_t m[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = map[o[k] >> kKeyShift];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[4 * u + k] = one(i + u * kStep + k, o[k], m[k]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<uint4*>(keys + i + u * kStep) = make_uint4(r[4 * u], r[4 * u + 1], r[4 * u + 2], r[4 * u + 3]);
    const uint64_t here = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
    for (int k = 0; k < 16; ++k) count_digit(wh[wave], (r[k] >> shift) & mask, here, true);
  }
  for (; i + 4 <= end; i += kStep) {
    const uint4 v = *reinterpret_cast<const uint4*>(keys + i);
    const uint32_t o[4] = {v.x, v.y, v.z, v.w};
    uint32_t m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = map[o[k] >> kKeyShift];
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = one(i + k, o[k], m[k]);
    *reinterpret_cast<uint4*>(keys + i) = make_uint4(r[0], r[1], r[2], r[3]);
    const uint64_t here = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
    for (int k = 0; k < 4; ++k) count_digit(wh[wave], (r[k] >> shift) & mask, here, true);
  }
  for (; i < end; ++i) {  // ragged tail (< 4 keys per lane)
    const uint32_t o = keys[i];
    const uint32_t r = one(i, o, map[o >> kKeyShift]);
    keys[i] = r;
    atomicAdd(&wh[wave][(r >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += kBlock) {
    uint32_t s2 = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s2 += wh[w][d];
    hist[(uint64_t)d * groups + blockIdx.x] = s2;
  }
}

// One workgroup per digit: exclusive scan of that digit's `groups` counters in place (groups <= 1024) and the
// digit's total. The scan across digits is folded into the downsweep prologue (256 values).
__global__ __launch_bounds__(256) void scan_kernel(uint32_t* __restrict__ hist, int groups,
                                                    uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_tot[4];
  uint32_t* row = hist + (uint64_t)blockIdx.x * groups;
  constexpr int kPerMax = (kMaxGroups + 255) / 256;
  const int per = (groups + 255) / 256;  // <= kPerMax
  const int begin = threadIdx.x * per;
  uint32_t v[kPerMax];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < kPerMax; ++i) v[i] = 0;
#pragma unroll
  for (int i = 0; i < kPerMax; ++i)
    if (i < per && begin + i < groups) {
      v[i] = row[begin + i];
      sum += v[i];
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    woff += (w < wave) ? wave_tot[w] : 0u;
    total += wave_tot[w];
  }
  uint32_t run = woff + inc - sum;
#pragma unroll
  for (int i = 0; i < kPerMax; ++i)
    if (i < per && begin + i < groups) {
      row[begin + i] = run;
      run += v[i];
    }
  if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// ---- shared pieces of the two downsweep kernels -------------------------------------------------

// R: digit values the pass can produce (256, or 128 for digits of <= 7 bits: 3 KB less LDS, which is what lets three
// workgroups of the 12-byte record kernel share a CU); thread t serves digit t, threads >= R only keep the barriers
template <int R = kRadix>
struct DigitState {
  uint32_t whist[kWaves][R];  // per-wave digit counters, then exclusive prefix over the waves
  uint32_t digit_base[R];     // global position of the next key of each digit for this workgroup
  uint32_t delta[R];          // digit_base - (digit's start inside the tile): LDS slot p goes to delta[digit] + p
  uint32_t wave_tot[kWaves];
};

// global base of digit t for this workgroup = (keys with a smaller digit) + (same digit, earlier workgroups)
template <int R>
__device__ __forceinline__ void init_digit_base(DigitState<R>& S, const uint32_t* __restrict__ offsets,
                                                const uint32_t* __restrict__ totals, int groups, int t, int lane, int wave) {
  const uint32_t tot = t < R ? totals[t] : 0u;  // kBlock == kRadix >= R
  uint32_t inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) S.wave_tot[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) woff += (w < wave) ? S.wave_tot[w] : 0u;
  if (t < R) {
    S.digit_base[t] = woff + inc - tot + offsets[(uint64_t)t * groups + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) S.whist[w][t] = 0;
  }
  __syncthreads();
}

// Rank of every key of this lane among the earlier keys of the same digit inside the wave's slice of the tile
// (stable: iteration-major, lane-minor == input order). Per key, 8 ballots build the mask of lanes holding the same
// digit; every lane reads the wave's digit counter, then the first lane of the group bumps it by the group size
// (non-returning LDS add). A wave's LDS operations retire in issue order, so the reads and adds of all kKpt
// iterations are issued back to back — no round trip per key — and every read still sees exactly the counts of the
// earlier iterations.
// The kernel is VALU-issue bound (a wave64 op takes 4 clocks on a 16-lane SIMD), so the mask arithmetic is written
// on 32-bit halves in the shape the ISA has single instructions for: one sign-extracting bit-field op per digit bit,
// one compare (the ballot), one three-input bit op per half (p & ~(ballot ^ m)), mbcnt for the lanes below.
template <int kKpt, typename KeyT, bool kFull, int R>
__device__ __forceinline__ void wave_rank_all(DigitState<R>& S, int wave, uint32_t wbase, uint32_t tile_n,
                                              const KeyT (&key)[kKpt], int shift, uint32_t mask, uint16_t (&lpos)[kKpt],
                                              int nbits = 8) {
  constexpr int kBatch = 8;  // adds in flight; more costs registers the 16-keys-per-lane kernel does not have
  static_assert(kKpt % kBatch == 0, "keys per lane must be a multiple of the batch");
#pragma unroll
  for (int i0 = 0; i0 < kKpt; i0 += kBatch) {
    uint32_t pre[kBatch], rank_in[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int i = i0 + j;
      const bool valid = kFull || wbase + i * 64 < tile_n;
      const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
      uint32_t plo = 0xffffffffu, phi = 0xffffffffu;
      if (!kFull) {
        const uint64_t vm = __ballot(valid);
        plo = (uint32_t)vm;
        phi = (uint32_t)(vm >> 32);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b >= 5 && b >= nbits) break;  // narrow digits (wave-uniform): the upper bits are zero in every lane
        int m;  // all ones when bit b of the digit is set (asm: keep the optimiser from re-deriving it the long way)
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(d), "n"(b));
        const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
        plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, (uint32_t)m, 0x90);  // p & ~(ballot ^ m)
        phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
      }
      rank_in[j] = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
      uint32_t* slot = &S.whist[wave][d];
      pre[j] = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (valid && rank_in[j] == 0)
        (void)__hip_atomic_fetch_add(slot, (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j) lpos[i0 + j] = (uint16_t)(pre[j] + rank_in[j]);
  }
}

// After all waves ranked their slices: per digit t the exclusive prefix over the waves (folded together with the
// digit's start inside the tile, so the LDS slot of a key is whist[wave][d] + its rank), the global position of the
// digit's run (delta) and the advance of digit_base. Starts and ends with a barrier.
template <int R>
__device__ __forceinline__ void digit_scan(DigitState<R>& S, int t, int lane, int wave) {
  __syncthreads();
  uint32_t pre[kWaves];
  uint32_t acc = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    pre[w] = acc;
    acc += t < R ? S.whist[w][t] : 0u;
  }
  uint32_t inc = acc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) S.wave_tot[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) woff += (w < wave) ? S.wave_tot[w] : 0u;
  const uint32_t start = woff + inc - acc;
  if (t < R) {
#pragma unroll
    for (int w = 0; w < kWaves; ++w) S.whist[w][t] = start + pre[w];
    const uint32_t base = S.digit_base[t];
    S.delta[t] = base - start;
    S.digit_base[t] = base + acc;
  }
  __syncthreads();
}

// ---- keys only ------------------------------------------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(kBlock, PCV_KEYS_WAVES) void downsweep_keys_kernel(const KeyT* __restrict__ keys_in,
                                                                   KeyT* __restrict__ keys_out, uint64_t n, uint64_t chunk,
                                                                   int groups, int shift, int nbits,
                                                                   const uint32_t* __restrict__ offsets,
                                                                   const uint32_t* __restrict__ totals) {
  constexpr int kKpt = kKptKeys, kTile = kBlock * kKpt;
  __shared__ KeyT skeys[kTile];
  __shared__ DigitState<kRadix> S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
  init_digit_base(S, offsets, totals, groups, t, lane, wave);

  const uint64_t begin = (uint64_t)blockIdx.x * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;
  const uint32_t wbase = wave * 64 * kKpt + lane;

  KeyT key[kKpt];
  {  // first tile
    const uint32_t tile_n = (uint32_t)((end - begin) < (uint64_t)kTile ? (end - begin) : (uint64_t)kTile);
#pragma unroll
    for (int i = 0; i < kKpt; ++i) {
      const uint32_t li = wbase + i * 64;
      key[i] = li < tile_n ? keys_in[begin + li] : (KeyT)0;
    }
  }
  for (uint64_t base = begin; base < end; base += kTile) {
    const uint32_t tile_n = (uint32_t)((end - base) < (uint64_t)kTile ? (end - base) : (uint64_t)kTile);
    uint16_t lpos[kKpt];
    if (tile_n == (uint32_t)kTile)
      wave_rank_all<kKpt, KeyT, true, kRadix>(S, wave, wbase, tile_n, key, shift, mask, lpos);
    else
      wave_rank_all<kKpt, KeyT, false, kRadix>(S, wave, wbase, tile_n, key, shift, mask, lpos);
    digit_scan(S, t, lane, wave);
#pragma unroll
    for (int i = 0; i < kKpt; ++i) {
      if (wbase + i * 64 < tile_n) {
        const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
        skeys[S.whist[wave][d] + lpos[i]] = key[i];
      }
    }
    // the key registers are free now: fetch the next tile while this one drains through LDS
    const uint64_t nbase = base + kTile;
    if (nbase < end) {
      const uint32_t next_n = (uint32_t)((end - nbase) < (uint64_t)kTile ? (end - nbase) : (uint64_t)kTile);
#pragma unroll
      for (int i = 0; i < kKpt; ++i) {
        const uint32_t li = wbase + i * 64;
        key[i] = li < next_n ? keys_in[nbase + li] : (KeyT)0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kKpt; ++j) {
      const uint32_t p = j * kBlock + t;
      if (p < tile_n) {
        const KeyT k = skeys[p];
        const uint32_t d = (uint32_t)(k >> shift) & mask;
        keys_out[S.delta[d] + p] = k;
      }
    }
#pragma unroll
    for (int w = 0; w < kWaves; ++w) S.whist[w][t] = 0;  // last read before the barrier above
    __syncthreads();
  }
}

// ---- records: u32 key + optional 16-byte payload + extra 4-byte planes -----------------------------
struct RecPtrs {
  const void* vec_in;  // may be null; uint4 (20-byte records) or uint2 (12-byte records) per key
  void* vec_out;
  int nplanes;          // extra u32 planes (0..8)
  const uint32_t* plane_in[8];
  uint32_t* plane_out[8];
};

// kPrefetch: the next tile's keys and payloads are loaded into the registers the LDS staging just freed, so that the
// loads are in flight while this tile drains through LDS to memory (as the keys-only kernel does).
#ifndef PCV_REC_WAVES
#define PCV_REC_WAVES 3
#endif
template <bool kHasVec, bool kPrefetch = false, typename VecT = uint4, int R = kRadix>
__global__ __launch_bounds__(kBlock, kHasVec ? PCV_REC_WAVES : 4) void downsweep_rec_kernel(const uint32_t* __restrict__ keys_in,
                                                                  uint32_t* __restrict__ keys_out, uint64_t n,
                                                                  uint64_t chunk, int groups, int shift, int nbits,
                                                                  const uint32_t* __restrict__ offsets,
                                                                  const uint32_t* __restrict__ totals, RecPtrs rp) {
  constexpr int kKpt = kKptRec, kTile = kBlock * kKpt;
  __shared__ uint32_t skeys[kTile];
  __shared__ VecT svec[kHasVec ? kTile : 1];
  __shared__ DigitState<R> S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
  init_digit_base(S, offsets, totals, groups, t, lane, wave);
  const VecT* __restrict__ vec_in = reinterpret_cast<const VecT*>(rp.vec_in);
  VecT* __restrict__ vec_out = reinterpret_cast<VecT*>(rp.vec_out);

  const uint64_t begin = (uint64_t)blockIdx.x * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;
  const uint32_t wbase = wave * 64 * kKpt + lane;

  uint32_t key[kKpt];
  VecT vec[kHasVec ? kKpt : 1];
  auto load_tile = [&](uint64_t base, uint32_t tile_n) {
#pragma unroll
    for (int i = 0; i < kKpt; ++i) {
      const uint32_t li = wbase + i * 64;
      const bool valid = li < tile_n;
      key[i] = valid ? keys_in[base + li] : 0u;
      if (kHasVec) vec[i] = valid ? vec_in[base + li] : VecT{};
    }
  };
  if (kPrefetch && begin < end) load_tile(begin, (uint32_t)((end - begin) < (uint64_t)kTile ? (end - begin) : (uint64_t)kTile));
  for (uint64_t base = begin; base < end; base += kTile) {
    const uint32_t tile_n = (uint32_t)((end - base) < (uint64_t)kTile ? (end - base) : (uint64_t)kTile);
    if (!kPrefetch) load_tile(base, tile_n);
    uint16_t lpos[kKpt];
    if (tile_n == (uint32_t)kTile)
      wave_rank_all<kKpt, uint32_t, true, R>(S, wave, wbase, tile_n, key, shift, mask, lpos, nbits);
    else
      wave_rank_all<kKpt, uint32_t, false, R>(S, wave, wbase, tile_n, key, shift, mask, lpos, nbits);
    digit_scan(S, t, lane, wave);
#pragma unroll
    for (int i = 0; i < kKpt; ++i) {
      if (wbase + i * 64 < tile_n) {
        const uint32_t d = (key[i] >> shift) & mask;
        const uint32_t p = S.whist[wave][d] + lpos[i];
        lpos[i] = (uint16_t)p;
        skeys[p] = key[i];
        if (kHasVec) svec[p] = vec[i];
      }
    }
    if (kPrefetch) {  // the key / payload registers are free: fetch the next tile while this one drains
      const uint64_t nbase = base + kTile;
      if (nbase < end) load_tile(nbase, (uint32_t)((end - nbase) < (uint64_t)kTile ? (end - nbase) : (uint64_t)kTile));
    }
    __syncthreads();
    uint32_t gidx[kKpt];
#pragma unroll
    for (int j = 0; j < kKpt; ++j) {
      const uint32_t p = j * kBlock + t;
      if (p < tile_n) {
        const uint32_t k = skeys[p];
        const uint32_t d = (k >> shift) & mask;
        const uint32_t g = S.delta[d] + p;
        gidx[j] = g;
        keys_out[g] = k;
        if (kHasVec) vec_out[g] = svec[p];
      }
    }
    for (int w = 0; w < rp.nplanes; ++w) {  // rare: intensity / Float64 high words / generic pairs API
      const uint32_t* __restrict__ src = rp.plane_in[w];
      uint32_t* __restrict__ dst = rp.plane_out[w];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kKpt; ++i) {
        const uint32_t li = wbase + i * 64;
        if (li < tile_n) skeys[lpos[i]] = src[base + li];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kKpt; ++j) {
        const uint32_t p = j * kBlock + t;
        if (p < tile_n) dst[gidx[j]] = skeys[p];
      }
    }
    if (t < R) {
#pragma unroll
      for (int w = 0; w < kWaves; ++w) S.whist[w][t] = 0;  // last read before the barrier after the LDS scatter
    }
    __syncthreads();
  }
}

// ---- 12-byte records, generalised geometry ---------------------------------------------------------------------------
// The same reduce-then-scan downsweep for the packed records of the single-chain build (u32 key + uint2 payload), with
// the workgroup size, the records per lane and the digit-state size as template parameters, so that the occupancy can
// be chosen: the 256-lane / 16-per-lane kernel above needs 235 VGPRs and 54 KB of LDS (two workgroups = 8 waves per CU).
//   BLOCK x KPT = tile (records staged through LDS per round); R = digit values (128 for digits of <= 7 bits);
//   WPE = waves per SIMD the register allocation is asked to admit; NT (unused): non-temporal stores cost 35 %.
// The output phase runs in groups of four LDS reads + four stores (a scheduling barrier between the groups keeps the
// compiler from hoisting all reads of the tile into registers), with the next tile's loads already in flight.
template <int NW, int R>
struct DigitStateN {
  uint32_t whist[NW][R];
  uint32_t digit_base[R];
  uint32_t delta[R];
  uint32_t wave_tot[NW];
};

// MAP (first pass of the single-chain build's record sort when the histogram came from the rank counts, hist_from_rows_kernel):
// the keys still carry PREDICTED leaf ranks; they are translated through the rank map (a copy in dynamic LDS) as they are
// loaded — rank := map[rank], and where the map flags a replayed leaf (bit 30) the record's first payload word becomes its
// input index — which is what upsweep_map_kernel does in a pass of its own otherwise.
// PL (round 5): ONE extra 4-byte plane travels with the record (the intensity of the reference binary's default payload,
// src/bin/build_octree.rs:47-52): 16 bytes per record through the same tiles, 32 KB more LDS.
// WC (round 5, experiment behind PCV_REC_WC in libpcv_hip_exp.so): whole-line write combining. A digit's records leave a tile
// only in 32-record blocks aligned to 32 records of the OUTPUT array (128 bytes of keys, 256 bytes of payloads); what is left
// of a digit's run (< 32 records) waits in a carry buffer in LDS for the next tile of the piece (tools/scatter_probe.hip: runs
// that start on 256-byte boundaries move the same bytes 20-26 % faster than runs at odd record offsets). 128 digit values only
// (48 KB of carry next to the 107 KB of the tile), no plane, no map copy in LDS.
// the settling pass's view of a leaf (one entry per digit value of the piece, in LDS)
struct alignas(16) FuseLeaf {  // 80 bytes: with the intensity plane two workgroups' tables, tiles and digit state fill the CU's LDS
  uint32_t lo, climb_base, flags;
  uint32_t count;      // points of the leaf as the HOST's tree has it: a record outside [lo, lo + count) is never settled here
  uint64_t xyz_off;    // byte offset of the leaf's .xyz content in the xyz blob
  uint64_t point_off;  // point offset of the leaf in the rgb / intensity blobs
  double mn[3], edge, inv_edge, inv_edge_lo;
};
static_assert(sizeof(FuseLeaf) == 80, "leaf table entry");
constexpr uint32_t kFuseSettles = 1u, kFuseU8 = 2u;
// FUSE (PcvSortFuse, pcv_internal.h): the pass is the LAST one of a two-pass sort whose pieces hold one value of the rank's lower
// digit each: a digit's run inside a tile is then ONE leaf's records at consecutive sorted slots. Waves take whole runs: the leaf's
// record comes through the scalar cache, and the run's records leave as final bytes / climber records (flagged leaves) or as
// 12-byte records like in the plain pass (the others: `settle` finishes those).
template <int BLOCK, int KPT, int R, int WPE, bool NT, int MAP, bool PL, bool WC, bool FUSE>
__device__ __forceinline__ void downsweep_rec12_body(const uint32_t* __restrict__ keys_in, uint32_t* __restrict__ keys_out, uint64_t n,
                                                     uint64_t chunk, int groups, int shift, int nbits, const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ totals, const uint2* __restrict__ vec_in,
                                                     uint2* __restrict__ vec_out, const uint32_t* __restrict__ gmap, uint32_t map_entries,
                                                     const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ plane_in, uint32_t* __restrict__ plane_out,
                                                     const PcvSortFuse& fuse, const uint8_t* __restrict__ color_in = nullptr,
                                                     uint32_t color_stride = 3) {
  constexpr int NW = BLOCK / 64, kTile = BLOCK * KPT, RW = R / 64;
  static_assert(!FUSE || (!WC && MAP == 0), "the settling pass: second pass of 12-byte records (+ the intensity plane)");
  __shared__ FuseLeaf sleaf[FUSE ? R : 1];  // FUSE: the leaf of digit value d in this piece: rank = d << low_bits | the piece's lower digit
  const uint32_t piece = order ? order[blockIdx.x] : blockIdx.x;
  static_assert(BLOCK >= R && R % 64 == 0 && KPT % 8 == 0, "geometry");
  __shared__ uint32_t skeys[kTile];
  __shared__ uint2 svec[kTile];
  __shared__ uint32_t splane[PL ? kTile : 1];
  static_assert(!WC || (R == 128 && BLOCK == 1024 && !PL && MAP != 1), "write combining: 128 digit values, 1 024 lanes, no plane, no LDS map");
  __shared__ uint32_t wc_key[WC ? R * 32 : 1];  // carry: the records of digit d at output positions [W_d, W_d + count_d)
  __shared__ uint2 wc_vec[WC ? R * 32 : 1];
  __shared__ uint32_t wc_count[WC ? R : 1], wc_W[WC ? R : 1], wc_gb[WC ? R : 1], wc_A[WC ? R : 1], wc_ts[WC ? R : 1], wc_cnt[WC ? R : 1];
  __shared__ uint32_t wc_seg[WC ? R + 1 : 1];  // exclusive prefix of the digits' block counts of this tile; [R] = their sum
  __shared__ DigitStateN<NW, R> S;
  extern __shared__ uint16_t smap_dyn[];  // MAP: map_entries half words: true rank (< 2^15) | replay mark << 15
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
  if (FUSE && t < R) {  // visible after the first barrier below
    const uint32_t r = ((uint32_t)t << fuse.low_bits) | (piece / fuse.blocks);
    FuseLeaf L{};
    if ((uint32_t)t <= mask && r < fuse.num_leaves) {
      const PcvNodeRec c = fuse.leaf_rec[r];
      L.lo = c.lo, L.climb_base = fuse.climb_base[r];
      L.flags = (fuse.leaf_fused[r] ? kFuseSettles : 0u) | (c.enc == PCV_ENC_UINT8 ? kFuseU8 : 0u);
      L.count = (r + 1u < fuse.num_leaves ? fuse.leaf_rec[r + 1u].lo : (uint32_t)n) - c.lo;  // leaves lie in rank order
      L.xyz_off = c.xyz_off, L.point_off = c.point_off;
      L.mn[0] = c.mn[0], L.mn[1] = c.mn[1], L.mn[2] = c.mn[2];
      L.edge = c.edge, L.inv_edge = c.inv_edge, L.inv_edge_lo = c.inv_edge_lo;
    }
    sleaf[t] = L;
  }
  if (MAP == 1)
    for (uint32_t i = t; i < map_entries; i += BLOCK) {  // visible after the first barrier below
      const uint32_t m = gmap[i];
      smap_dyn[i] = (uint16_t)((m & 0x7fffu) | (((m >> 30) & 1u) << 15));
    }
  {  // global base of digit t for this workgroup = (keys with a smaller digit) + (same digit, earlier workgroups)
    const uint32_t tot = t < R ? totals[t] : 0u;
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63 && wave < RW) S.wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < RW; ++w) woff += (w < wave) ? S.wave_tot[w] : 0u;
    if (t < R) S.digit_base[t] = woff + inc - tot + offsets[(uint64_t)t * groups + piece];
    if (WC && t < R) wc_count[t] = 0;
    for (int k = t; k < NW * R; k += BLOCK) (&S.whist[0][0])[k] = 0;
    __syncthreads();
  }
  uint64_t begin = (uint64_t)piece * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;
  if (ranges) {
    begin = ranges[piece].x;
    end = ranges[piece].y;
  }
  const uint32_t wbase = wave * 64 * KPT + lane;

  uint32_t key[KPT];
  uint2 vec[KPT];
  uint32_t pln[PL ? KPT : 1];
  auto load_tile = [&](uint64_t base, uint32_t tile_n) {
    const uint32_t* __restrict__ kp = keys_in + base + wbase;
    const uint2* __restrict__ vp = vec_in + base + wbase;
    const uint32_t* __restrict__ pp = PL ? plane_in + base + wbase : nullptr;
    if (tile_n == (uint32_t)kTile) {  // full tile: straight-line loads off one base address each
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        key[i] = kp[i * 64];
        vec[i] = vp[i * 64];
        if (PL) pln[i] = pp[i * 64];
      }
    } else {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const bool valid = wbase + i * 64 < tile_n;
        key[i] = valid ? kp[i * 64] : 0u;
        vec[i] = valid ? vp[i * 64] : make_uint2(0u, 0u);
        if (PL) pln[i] = valid ? pp[i * 64] : 0u;
      }
    }
    if (MAP != 0 && color_in) {  // (wave-uniform) the first pass of records that left the chain pass without their colour
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const uint64_t idx = base + wbase + (uint32_t)(i * 64);
        if (tile_n == (uint32_t)kTile || wbase + i * 64 < tile_n) {
          const uint8_t* c = color_in + idx * color_stride;
          uint32_t rgb;  // r | g << 8 | b << 16 (pcv_load_rgb: one unaligned dword where a fourth byte exists behind the colour)
          if (idx + 1 < n) {
            __builtin_memcpy(&rgb, c, 4);
            rgb &= 0xffffffu;
          } else {
            rgb = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
          }
          key[i] |= rgb >> 16;
          vec[i].y |= (rgb & 0xffffu) << 16;
        }
      }
    }
    if (MAP == 1) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const uint32_t pr = key[i] >> 8;
        const uint32_t m = smap_dyn[pr < map_entries ? pr : 0u];
        if (__builtin_expect((m & 0x8000u) != 0u, 0)) vec[i].x = (uint32_t)(base + wbase + i * 64);  // replay: the input index
        key[i] = ((m & 0x7fffu) << 8) | (key[i] & 0xffu);
      }
    }
    if (MAP == 2) {  // trees of more than 16 384 predicted nodes: the map (4 bytes per node, L2-resident) is gathered as it is
      uint32_t m[KPT];
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const uint32_t pr = key[i] >> 8;
        m[i] = gmap[pr < map_entries ? pr : 0u];
      }
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (__builtin_expect((m[i] & (1u << 30)) != 0u, 0)) vec[i].x = (uint32_t)(base + wbase + i * 64);  // replay: the input index
        key[i] = ((m[i] & PCV_SPEC_INDEX_MASK_SORT) << 8) | (key[i] & 0xffu);
      }
    }
  };
  // (MAP: the copy of the map above is complete: the digit-base prologue ended with a barrier)
  if (begin < end) load_tile(begin, (uint32_t)((end - begin) < (uint64_t)kTile ? (end - begin) : (uint64_t)kTile));
  for (uint64_t base = begin; base < end; base += kTile) {
    const uint32_t tile_n = (uint32_t)((end - base) < (uint64_t)kTile ? (end - base) : (uint64_t)kTile);
    const bool full = tile_n == (uint32_t)kTile;
    // rank of every record among the earlier records of the same digit inside the wave's slice (see wave_rank_all)
    uint16_t lpos[KPT];
#pragma unroll
    for (int i0 = 0; i0 < KPT; i0 += 8) {
      uint32_t pre[8], rank_in[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j;
        const bool valid = full || wbase + i * 64 < tile_n;
        const uint32_t d = (key[i] >> shift) & mask;
        uint32_t plo = 0xffffffffu, phi = 0xffffffffu;
        if (!full) {
          const uint64_t vm = __ballot(valid);
          plo = (uint32_t)vm;
          phi = (uint32_t)(vm >> 32);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          if (b >= 5 && b >= nbits) break;
          int m;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(d), "n"(b));
          const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
          plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, (uint32_t)m, 0x90);
          phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
        }
        rank_in[j] = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        uint32_t* slot = &S.whist[wave][d];
        pre[j] = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (valid && rank_in[j] == 0)
          (void)__hip_atomic_fetch_add(slot, (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) lpos[i0 + j] = (uint16_t)(pre[j] + rank_in[j]);
    }
    // per digit: exclusive prefix over the waves + the digit's start inside the tile; global position of the digit's run
    __syncthreads();
    {
      uint32_t pre[NW];
      uint32_t acc = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        pre[w] = acc;
        acc += t < R ? S.whist[w][t] : 0u;
      }
      // WC: the digit's aligned blocks of this tile are scanned in the upper half word of the same prefix (<= 8 192 records
      // and <= 384 blocks per tile)
      uint32_t wcW = 0, wcA = 0, nseg = 0, gb0 = 0;
      if (WC && t < R) {
        gb0 = S.digit_base[t];
        wcW = gb0 - wc_count[t];
        wcA = (gb0 + acc) & ~31u;
        nseg = wcA > wcW ? (wcA >> 5) - (wcW >> 5) : 0u;
      }
      const uint32_t val = WC ? (acc | (nseg << 16)) : acc;
      uint32_t inc = val;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63 && wave < RW) S.wave_tot[wave] = inc;
      __syncthreads();
      uint32_t woff = 0;
#pragma unroll
      for (int w = 0; w < RW; ++w) woff += (w < wave) ? S.wave_tot[w] : 0u;
      const uint32_t excl = woff + inc - val;
      const uint32_t start = WC ? (excl & 0xffffu) : excl;
      if (t < R) {
#pragma unroll
        for (int w = 0; w < NW; ++w) S.whist[w][t] = start + pre[w];
        const uint32_t gb = S.digit_base[t];
        S.delta[t] = gb - start;
        S.digit_base[t] = gb + acc;
        if (WC) {
          wc_W[t] = wcW, wc_gb[t] = gb0, wc_A[t] = wcA, wc_ts[t] = start, wc_cnt[t] = acc;
          wc_seg[t] = excl >> 16;
          if (t == R - 1) wc_seg[R] = (excl >> 16) + nseg;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (full || wbase + i * 64 < tile_n) {
        const uint32_t d = (key[i] >> shift) & mask;
        const uint32_t p = S.whist[wave][d] + lpos[i];
        skeys[p] = key[i];
        svec[p] = vec[i];
        if (PL) splane[p] = pln[i];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // the loads below must not be hoisted over the LDS scatter (twice the registers)
    {  // the key / payload registers are free: fetch the next tile while this one drains
      const uint64_t nbase = base + kTile;
      if (nbase < end) load_tile(nbase, (uint32_t)((end - nbase) < (uint64_t)kTile ? (end - nbase) : (uint64_t)kTile));
    }
    __syncthreads();
    if constexpr (WC) {
      // every half wave writes whole 32-record blocks: 128 contiguous, aligned bytes of keys and 256 of payloads
      const uint32_t ns = wc_seg[R];
      const uint32_t l32 = (uint32_t)t & 31u;
      for (uint32_t q = (uint32_t)t >> 5; q < ns; q += BLOCK / 32) {
        uint32_t lo = 0, hi = R;  // the digit of block q: the last one whose prefix is <= q
#pragma unroll
        for (int it = 0; it < 7; ++it) {
          const uint32_t mid = (lo + hi) >> 1;
          if (wc_seg[mid] <= q) lo = mid;
          else hi = mid;
        }
        const uint32_t W = wc_W[lo], gb = wc_gb[lo];
        const uint32_t P = (((W >> 5) + (q - wc_seg[lo])) << 5) + l32;
        if (P >= W) {  // (the first block of a piece's digit may begin inside a line)
          uint32_t k;
          uint2 v;
          if (P < gb) {
            k = wc_key[lo * 32 + (P - W)];
            v = wc_vec[lo * 32 + (P - W)];
          } else {
            const uint32_t src = wc_ts[lo] + (P - gb);
            k = skeys[src];
            v = svec[src];
          }
          keys_out[P] = k;
          vec_out[P] = v;
        }
      }
      __syncthreads();
      {  // what is left of every digit's run goes to (or stays in) its carry: lanes 8 d .. 8 d + 7 serve digit d
        const uint32_t d = (uint32_t)t >> 3;
        const uint32_t c = wc_count[d], W = wc_W[d], gb = wc_gb[d], A = wc_A[d], cnt = wc_cnt[d], ts = wc_ts[d];
        const bool wrote = A > W;
        const uint32_t newc = wrote ? gb + cnt - A : c + cnt;
#pragma unroll
        for (uint32_t k8 = 0; k8 < 4; ++k8) {
          const uint32_t sl = ((uint32_t)t & 7u) + 8u * k8;
          if (wrote ? sl < newc : (sl >= c && sl < newc)) {
            const uint32_t src = wrote ? ts + (A - gb) + sl : ts + (sl - c);
            wc_key[d * 32 + sl] = skeys[src];
            wc_vec[d * 32 + sl] = svec[src];
          }
        }
        if (((uint32_t)t & 7u) == 0u) wc_count[d] = newc;
      }
    } else if constexpr (FUSE) {
      // every lane finishes the records at its own tile positions (consecutive lanes = consecutive sorted slots of a run): the run's
      // leaf comes out of the piece's table in LDS (lanes of one run read one address: a broadcast)
#pragma unroll 2
      for (int i = 0; i < KPT; ++i) {
        const uint32_t p = (uint32_t)i * BLOCK + (uint32_t)t;
        if (!(full || p < tile_n)) continue;
        const uint32_t k = skeys[p];
        const uint2 q = svec[p];
        const uint32_t d = (k >> shift) & mask;
        const FuseLeaf& L = sleaf[d];
        const uint32_t g = S.delta[d] + p;  // sorted slot
        const uint32_t flags = L.flags;
        const uint32_t inten = PL ? splane[p] : 0u;
        // (g - lo < count always holds when the device's rank map and the host's tree agree — the build checks that they do,
        // but only at its end: a record that falls outside its leaf keeps its 12 bytes instead of being written anywhere)
        if (!(flags & kFuseSettles) || g - L.lo >= L.count) {  // `settle` finishes this leaf: the record as in the plain pass
          keys_out[g] = k;
          vec_out[g] = q;
          if (PL) plane_out[g] = inten;
          continue;
        }
        const uint32_t j = g - L.lo;  // position in the leaf's stream
        const uint32_t rgb = (q.y >> 16) | ((k & 0xffu) << 16);
        const uint32_t c0 = q.x & 0xffffu, c1 = q.x >> 16, c2 = q.y & 0xffffu;
        if ((j & 7u) == 0) {  // every eighth point climbs: its record for `climb`, dense per leaf
          if (PL)  // (the leaf's rank: this digit above the piece's lower digit)
            reinterpret_cast<PcvClimber*>(fuse.climbers)[L.climb_base + (j >> 3)] =
                PcvClimber{make_uint4(c0, c1, c2, rgb), (d << fuse.low_bits) | (piece / fuse.blocks), g, inten, 0u};
          else reinterpret_cast<uint4*>(fuse.climbers)[L.climb_base + (j >> 3)] = make_uint4(c0, c1, c2, rgb);
          continue;
        }
        // final rewrite encode(decode(code)) at the leaf's own level (promote_final, pcv_settle_dev.h) with the
        // encoding's constants as per-lane values: the same operations in the same order for u8 and u16
        const bool u8 = (flags & kFuseU8) != 0;
        const double maxval = u8 ? 255.0 : 65535.0;
        const PcvRecip rm = u8 ? PCV_RECIP_255 : PCV_RECIP_65535;
        uint32_t out[3];
        const uint32_t cin[3] = {c0, c1, c2};
#if PCV_FUSE_DIAG == 2
        out[0] = c0, out[1] = c1, out[2] = c2;
        if (L.inv_edge == 123.0) {
#else
        if (__builtin_expect(L.inv_edge != 0.0, 1)) {
#endif
          const PcvRecip ie{L.inv_edge, L.inv_edge_lo};
#pragma unroll
          for (int a = 0; a < 3; ++a)
            out[a] = pcv_fix_encode<false>(__fma_rn(pcv_div_code((double)cin[a], rm), L.edge, L.mn[a]), L.mn[a], L.edge, ie, maxval);
#if PCV_FUSE_DIAG == 2
        } else if (L.inv_edge == 124.0) {
#else
        } else {
#endif
          const uint32_t enc = u8 ? PCV_ENC_UINT8 : PCV_ENC_UINT16;
#pragma unroll
          for (int a = 0; a < 3; ++a)
            out[a] = (uint32_t)pcv_encode_coord(enc, pcv_decode_coord(enc, cin[a], L.mn[a], L.edge), L.mn[a], L.edge, PcvRecip{0.0, 0.0});
        }
        const uint32_t slot = j - (j >> 3) - 1u;
        const bool odd = (slot & 1u) != 0;
#if PCV_FUSE_DIAG == 1  // (timing experiments, tools/build_variants.sh: 1 = no final stores, 2 = no rewrite; never shipped)
        if (out[0] != 0x7fffffffu) continue;
#endif
        const uint64_t pidx = L.point_off + slot;
        if (PL) reinterpret_cast<uint32_t*>(fuse.inten_blob)[pidx] = inten;
        // 3 bytes at 3 x pidx: one 2-byte store at the EVEN address of the three + one byte. The parity is the address's, i.e.
        // pidx's, not slot's: a leaf's point_off may be odd (the .xyz stores below may use slot's, xyz_off is
        // 16-byte aligned)
        uint8_t* cd = fuse.rgb_blob + pidx * 3;
        const bool odd_rgb = (pidx & 1u) != 0;
        *reinterpret_cast<uint16_t*>(cd + (odd_rgb ? 1 : 0)) = (uint16_t)(odd_rgb ? rgb >> 8 : rgb);
        cd[odd_rgb ? 0 : 2] = (uint8_t)(odd_rgb ? rgb : rgb >> 16);
        if (u8) {
          uint8_t* x = fuse.xyz_blob + L.xyz_off + (uint64_t)slot * 3;
          *reinterpret_cast<uint16_t*>(x + (odd ? 1 : 0)) = (uint16_t)(odd ? out[1] | (out[2] << 8) : out[0] | (out[1] << 8));
          x[odd ? 0 : 2] = (uint8_t)(odd ? out[0] : out[2]);
        } else {  // 6 bytes at 6 x slot: one 4-byte store at the 4-aligned address of the six + one 2-byte store
          uint8_t* x = fuse.xyz_blob + L.xyz_off + (uint64_t)slot * 6;
          *reinterpret_cast<uint32_t*>(x + (odd ? 2 : 0)) = odd ? out[1] | (out[2] << 16) : out[0] | (out[1] << 16);
          *reinterpret_cast<uint16_t*>(x + (odd ? 0 : 4)) = (uint16_t)(odd ? out[0] : out[2]);
        }
      }
    } else {
#pragma unroll
    for (int j0 = 0; j0 < KPT; j0 += 4) {
      uint32_t k4[4];
      uint2 v4[4];
      uint32_t p4[PL ? 4 : 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t p = (j0 + j) * BLOCK + t;
        k4[j] = skeys[p];
        v4[j] = svec[p];
        if (PL) p4[j] = splane[p];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t p = (j0 + j) * BLOCK + t;
        if (full || p < tile_n) {
          const uint32_t g = S.delta[(k4[j] >> shift) & mask] + p;
          keys_out[g] = k4[j];
          vec_out[g] = v4[j];
          if (PL) plane_out[g] = p4[j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    for (int k = t; k < NW * R; k += BLOCK) (&S.whist[0][0])[k] = 0;  // last read before the barrier above
    __syncthreads();
  }
  if constexpr (WC) {  // end of the piece: the carries leave as they are (the tail of every digit's run)
    const uint32_t d = (uint32_t)t >> 3;
    const uint32_t c = wc_count[d], W = S.digit_base[d] - c;
#pragma unroll
    for (uint32_t k8 = 0; k8 < 4; ++k8) {
      const uint32_t sl = ((uint32_t)t & 7u) + 8u * k8;
      if (sl < c) {
        keys_out[W + sl] = wc_key[d * 32 + sl];
        vec_out[W + sl] = wc_vec[d * 32 + sl];
      }
    }
  }
}


template <int BLOCK, int KPT, int R, int WPE, bool NT, int MAP = 0 /* 1: the map in LDS (half words), 2: in global memory */, bool PL = false,
          bool WC = false>
__global__ __launch_bounds__(BLOCK, WPE) void downsweep_rec12_kernel(const uint32_t* __restrict__ keys_in,
                                                                     uint32_t* __restrict__ keys_out, uint64_t n, uint64_t chunk,
                                                                     int groups, int shift, int nbits,
                                                                     const uint32_t* __restrict__ offsets,
                                                                     const uint32_t* __restrict__ totals,
                                                                     const uint2* __restrict__ vec_in, uint2* __restrict__ vec_out,
                                                                     const uint32_t* __restrict__ gmap = nullptr, uint32_t map_entries = 0,
                                                                     const uint2* __restrict__ ranges = nullptr /* set: piece k
                                                                     = the records [ranges[k].x, ranges[k].y) instead of chunk k */,
                                                                     const uint32_t* __restrict__ order = nullptr /* set: workgroup
                                                                     b takes piece order[b] (largest pieces first) */,
                                                                     const uint32_t* __restrict__ plane_in = nullptr,
                                                                     uint32_t* __restrict__ plane_out = nullptr,
                                                                     const uint8_t* __restrict__ color_in = nullptr /* MAP != 0: the
                                                                     records come without colour, record i's is here */,
                                                                     uint32_t color_stride = 3) {
  downsweep_rec12_body<BLOCK, KPT, R, WPE, NT, MAP, PL, WC, false>(keys_in, keys_out, n, chunk, groups, shift, nbits, offsets, totals, vec_in,
                                                                  vec_out, gmap, map_entries, ranges, order, plane_in, plane_out, PcvSortFuse(),
                                                                  color_in, color_stride);
}
// the settling form of the second pass (FUSE above): a kernel of its own name for the profiles
// (R = 128 digit values: the second digit of a rank of <= 15 bits has <= 7 bits; 256 for ranks of 16 bits)
template <bool PL, int BLOCK = 1024, int R = 128>
__global__ __launch_bounds__(BLOCK, 4) void downsweep_settle_kernel(const uint32_t* __restrict__ keys_in, uint32_t* __restrict__ keys_out,
                                                                   uint64_t n, uint64_t chunk, int groups, int shift, int nbits,
                                                                   const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ totals,
                                                                   const uint2* __restrict__ vec_in, uint2* __restrict__ vec_out,
                                                                   const uint2* __restrict__ ranges, const uint32_t* __restrict__ order,
                                                                   const uint32_t* __restrict__ plane_in, uint32_t* __restrict__ plane_out,
                                                                   PcvSortFuse fuse) {
  downsweep_rec12_body<BLOCK, 8, R, 4, false, 0, PL, false, true>(keys_in, keys_out, n, chunk, groups, shift, nbits, offsets, totals, vec_in,
                                                                   vec_out, nullptr, 0u, ranges, order, plane_in, plane_out, fuse);
}

// First-pass histogram of the record sort from the per-workgroup rank counts (rank_hist rows, pcv_encode.hip) and the rank map:
// workgroup g's count of digit d = sum over the predicted leaves b whose TRUE rank has digit d of rows[g][b]. One workgroup
// per sort workgroup; the keys are not read.
__global__ __launch_bounds__(256) void hist_from_rows_kernel(const uint32_t* __restrict__ rows, uint32_t nbins,
                                                              const uint32_t* __restrict__ map, int rank_shift, uint32_t mask, int groups,
                                                              uint32_t* __restrict__ hist /* [digit][groups] */) {
  __shared__ uint32_t h[kRadix];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t* row = rows + (uint64_t)blockIdx.x * nbins;
  for (uint32_t b = threadIdx.x; b < nbins; b += 256) {
    const uint32_t c = row[b];
    if (c) atomicAdd(&h[((map[b] & PCV_SPEC_INDEX_MASK_SORT) >> rank_shift) & mask], c);
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * groups + blockIdx.x] = threadIdx.x <= mask ? h[threadIdx.x] : 0u;  // all kRadix rows are scanned
}

// Two-pass sorts: BOTH histograms from the rank counts. Per sort workgroup g the counts are re-indexed by TRUE rank, lower
// digit major — tr[d1 * D2 + d2] — and kept (rows_true[g][.]); the first pass's histogram is their sum over d2. The first
// pass writes its output ordered by (d1, g); the second pass cuts THAT sequence into pieces of whole runs — piece k = digit
// d1 = k / blocks, workgroups [blk * gpb, (blk + 1) * gpb) of the first pass — so a piece's digit counts are sums of
// rows_true entries and its record range follows from the first pass's offsets: the second pass needs no counting pass over
// the keys either (pass2_layout_kernel). Pieces differ in size (by the popularity of d1) instead of being equal chunks.
__global__ __launch_bounds__(256) void hist12_from_rows_kernel(const uint32_t* __restrict__ rows, uint32_t nbins,
                                                                const uint32_t* __restrict__ map, int nbits1, int nbits2, int groups,
                                                                uint32_t* __restrict__ hist1 /* [d1][groups] */,
                                                                uint32_t* __restrict__ rows_true /* [groups][D1 * D2] */,
                                                                int msd /* experiments: the FIRST pass takes the rank's upper nbits1 bits */) {
  extern __shared__ uint32_t tr[];  // D1 x D2 counters: 64 KB for ranks of 14 bits, 128 KB for 15; 16 bits: two rounds of 128 KB
  const uint32_t D1 = 1u << nbits1, D2 = 1u << nbits2, TB = D1 * D2;
  const int sh1 = msd ? nbits2 : 0, sh2 = msd ? 0 : nbits1;
  const uint32_t rounds = TB > 32768u ? TB / 32768u : 1u, D1r = D1 / rounds, TBr = D1r * D2;  // a round takes D1r values of the first digit
  const uint32_t* row = rows + (uint64_t)blockIdx.x * nbins;
  uint32_t* out = rows_true + (uint64_t)blockIdx.x * TB;
  hist1[(uint64_t)threadIdx.x * groups + blockIdx.x] = 0;  // all kRadix rows are scanned (digit values >= D1 stay empty)
  for (uint32_t rd = 0; rd < rounds; ++rd) {
    const uint32_t d1_lo = rd * D1r;
    for (uint32_t i = threadIdx.x; i < TBr; i += 256) tr[i] = 0;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += 256) {
      const uint32_t c = row[b];
      if (c) {
        const uint32_t r = map[b] & PCV_SPEC_INDEX_MASK_SORT;
        const uint32_t d1 = ((r >> sh1) & (D1 - 1u)) - d1_lo;
        if (d1 < D1r) atomicAdd(&tr[d1 * D2 + ((r >> sh2) & (D2 - 1u))], c);
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < TBr; i += 256) out[(uint64_t)d1_lo * D2 + i] = tr[i];
    if (threadIdx.x < D1r) {
      uint32_t s = 0;
      for (uint32_t d2 = 0; d2 < D2; ++d2) s += tr[threadIdx.x * D2 + ((d2 + threadIdx.x) & (D2 - 1u))];  // skewed: no bank conflict
      hist1[(uint64_t)(d1_lo + threadIdx.x) * groups + blockIdx.x] = s;
    }
    __syncthreads();
  }
}
// piece k of the second pass (see above): its digit counts and its record range. offsets1 / totals1: the first pass's scanned
// histogram. One workgroup of 256 lanes per piece.
__global__ __launch_bounds__(256) void pass2_layout_kernel(const uint32_t* __restrict__ rows_true, int nbits1, int nbits2, int groups,
                                                            int blocks, int gpb, const uint32_t* __restrict__ offsets1,
                                                            const uint32_t* __restrict__ totals1, int pieces,
                                                            uint32_t* __restrict__ hist2 /* [d2][pieces] */, uint2* __restrict__ ranges,
                                                            uint32_t* __restrict__ order /* pieces by falling size of their digit */) {
  __shared__ uint32_t part[4][256];
  const uint32_t D1 = 1u << nbits1, D2 = 1u << nbits2, TB = D1 * D2;
  const int k = blockIdx.x, d1 = k / blocks, blk = k % blocks;
  const int g_lo = blk * gpb, g_hi = (g_lo + gpb < groups) ? g_lo + gpb : groups;
  const uint32_t d2 = threadIdx.x & (D2 - 1u), lanes_per_g = 256u / D2, sub = threadIdx.x / D2;  // D2 <= 256
  uint32_t s = 0;
  for (int g = g_lo + (int)sub; g < g_hi; g += (int)lanes_per_g) s += rows_true[(uint64_t)g * TB + (uint32_t)d1 * D2 + d2];
  (&part[0][0])[threadIdx.x] = s;
  __syncthreads();
  {
    uint32_t tot = 0;
    if (threadIdx.x < D2)
      for (uint32_t q = 0; q < lanes_per_g; ++q) tot += (&part[0][0])[q * D2 + threadIdx.x];
    hist2[(uint64_t)threadIdx.x * pieces + k] = tot;  // all kRadix rows are scanned
  }
  if (threadIdx.x == 0) {
    uint32_t start = 0;  // records with a smaller first digit
    for (int d = 0; d < d1; ++d) start += totals1[d];
    const uint32_t b = g_lo < groups ? start + offsets1[(uint64_t)d1 * groups + g_lo] : start + totals1[d1];
    const uint32_t e = g_hi < groups ? start + offsets1[(uint64_t)d1 * groups + g_hi] : start + totals1[d1];
    ranges[k] = make_uint2(b, e);
    // launch order: the pieces of the most popular first digits first (a piece's size follows its digit's total)
    const uint32_t mine = totals1[d1];
    uint32_t before = 0;
    for (int d = 0; d < (int)D1; ++d) {
      const uint32_t o = totals1[d];
      before += (o > mine || (o == mine && d < d1)) ? 1u : 0u;
    }
    order[before * (uint32_t)blocks + (uint32_t)blk] = (uint32_t)k;
  }
}

// (experiments, PCV_SORT_MSD) Most significant digit FIRST: the second pass then sorts every bucket of the first one by the lower
// digit INSIDE the bucket's own range of the output. hist2[d2][piece] (piece = (d1, blk)) turns into the absolute position of
// that (digit, piece) run: start of bucket d1 + records of the bucket with a smaller d2 + same d2, earlier blocks; the
// downsweep's digit prefix (totals) is zeroed. One workgroup per first digit.
__global__ __launch_bounds__(256) void msd_offsets_kernel(uint32_t* __restrict__ hist2, int pieces, int blocks, int nbits2,
                                                           const uint32_t* __restrict__ totals1, uint32_t* __restrict__ totals2) {
  __shared__ uint32_t wave_tot[4];
  const int d1 = blockIdx.x;
  const uint32_t D2 = 1u << nbits2, total = D2 * (uint32_t)blocks;
  if (d1 == 0) totals2[threadIdx.x] = 0;  // kRadix == 256 entries
  uint32_t start = 0;
  for (int d = 0; d < d1; ++d) start += totals1[d];
  const uint32_t per = (total + 255u) / 256u;
  uint32_t running = start;  // (uniform) everything before the chunk of 256 x per values in flight
  // value i of the bucket's sequence: digit i / blocks, block i % blocks
  auto at = [&](uint32_t i) -> uint32_t& { return hist2[(uint64_t)(i / (uint32_t)blocks) * pieces + (uint32_t)d1 * blocks + i % (uint32_t)blocks]; };
  const uint32_t begin = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < per; ++i)
    if (begin + i < total) sum += at(begin + i);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) woff += (w < wave) ? wave_tot[w] : 0u;
  uint32_t run = running + woff + inc - sum;
  for (uint32_t i = 0; i < per; ++i)
    if (begin + i < total) {
      uint32_t& v = at(begin + i);
      const uint32_t c = v;
      v = run;
      run += c;
    }
}

// Geometry of the 12-byte record downsweep. What moves this kernel is the length of the write runs (tile / digit
// values), not the occupancy (r03a / r03d A/B at 100 M records, both passes together, one box per line):
//   tiles of 4 096: 256 lanes x 16 at 8 waves per CU 1.36-1.40 ms, at 12 waves per CU 1.40, 512 x 8 at 16 waves per CU 1.36
//   tiles of 8 192: 1 024 x 8 1.20 ms (ships), 512 x 16 1.25; non-temporal stores +35 %
//   tiles of 16 384 by staging keys and payloads one after the other through the same LDS: 1.27-1.29 against 1.01-1.11
//   (two more barriers per tile, one workgroup per CU) — dropped
// PCV_REC_VARIANT (libpcv_hip_exp.so): 3 = 1 024 x 8 (default), 4 = 512 x 16, 2 = 512 x 8 (tiles of 4 096), 0 = the
// 256-lane kernel above
static void rec12_launch(pcv_ctx* ctx, int variant, const SortGeom& g, const uint32_t* src, uint32_t* dst, uint64_t n, int shift,
                         int nbits, const uint32_t* hist, const uint32_t* totals, const uint2* vin, uint2* vout,
                         const uint32_t* pin = nullptr, uint32_t* pout = nullptr) {
#define PCV_REC12(B, K, R, W, NT)                                                                                              \
  hipLaunchKernelGGL((downsweep_rec12_kernel<B, K, R, W, NT>), dim3(g.groups), dim3(B), 0, ctx->stream, src, dst, n, g.chunk, g.groups, \
                     shift, nbits, hist, totals, vin, vout)
  const bool narrow = nbits <= 7;
  if (pin) {  // records with one plane: tiles of 8 192 only
    if (narrow)
      hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, 128, 4, false, 0, true>), dim3(g.groups), dim3(1024), 0, ctx->stream, src, dst, n, g.chunk,
                         g.groups, shift, nbits, hist, totals, vin, vout, (const uint32_t*)nullptr, 0u, (const uint2*)nullptr,
                         (const uint32_t*)nullptr, pin, pout);
    else
      hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, 256, 4, false, 0, true>), dim3(g.groups), dim3(1024), 0, ctx->stream, src, dst, n, g.chunk,
                         g.groups, shift, nbits, hist, totals, vin, vout, (const uint32_t*)nullptr, 0u, (const uint2*)nullptr,
                         (const uint32_t*)nullptr, pin, pout);
    return;
  }
  switch (variant) {
    case 2:
      if (narrow) PCV_REC12(512, 8, 128, 4, false);
      else PCV_REC12(512, 8, 256, 4, false);
      break;
    case 4:
      if (narrow) PCV_REC12(512, 16, 128, 2, false);
      else PCV_REC12(512, 16, 256, 2, false);
      break;
    default:
      if (narrow) PCV_REC12(1024, 8, 128, 4, false);
      else PCV_REC12(1024, 8, 256, 4, false);
  }
#undef PCV_REC12
}

// true-rank counters per sort workgroup the scratch holds (hist12_from_rows_kernel): 2^14, and 2^15 for clouds big enough to have
// that many leaves (128 MB of scratch instead of 64), 2^16 from 500 M points on (256 MB)
static uint32_t rows_true_bins(uint64_t n) {
#ifdef PCV_EXPERIMENTS
  // PCV_ROWS_TRUE_BINS=32768 / 65536 (libpcv_hip_exp.so): the 15- / 16-bit rank geometries on a cloud small enough for the CPU
  // oracle to check them (tests/test_gpu_single_chain.py)
  static const uint32_t forced = [] {
    const char* e = pcv_experiment("PCV_ROWS_TRUE_BINS");
    return e ? (uint32_t)atoi(e) : 0u;
  }();
  if (forced) return forced;
#endif
  return n >= 500000000ull ? 65536u : n >= 200000000ull ? 32768u : 16384u;
}

template <typename KeyT>
int radix_sort(pcv_ctx* ctx, KeyT* a, KeyT* b, uint64_t n, int begin_bit, int end_bit, PcvSortPayload* payload,
               void* scratch, bool* result_in_a, const uint32_t* map = nullptr, uint32_t map_entries = 0,
               const uint32_t* rows = nullptr, PcvSortSecond* second = nullptr) {
  *result_in_a = true;
  if (second) second->pending = false;
  if (n == 0 || end_bit <= begin_bit) return PCV_OK;
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "radix sort: n must be < 2^32 - 1");
  const bool records = payload && (payload->vec_in || payload->nwords > 0);
  if (records && sizeof(KeyT) != 4) return ctx->fail(PCV_E_INVALID, "record sort needs 32-bit keys");
  const bool compact = records && payload->vec_in && payload->vec_bytes == 8;  // 12-byte records
  // geometry of the 12-byte record downsweep: 1 024 lanes x 8 records, tiles of 8 192 (rec12_launch)
  static const int rec_variant = [] {
    const char* e = pcv_experiment("PCV_REC_VARIANT");
    return e ? atoi(e) : 3;
  }();
  // 12-byte records, alone or with ONE 4-byte plane (intensity); more planes (the exact pipeline's wide codes) take the 256-lane kernel
  const bool with_plane = compact && payload->nwords == 1 && rec_variant == 3;
  const bool rec12 = compact && (payload->nwords == 0 || with_plane) && rec_variant > 0;
  SortGeom g = make_geom(n, rec12 && rec_variant != 2 ? 8192 : kTileUnit);
  uint32_t* hist = (uint32_t*)scratch;
  uint32_t* totals = hist + (size_t)kRadix * kMaxGroups;
  bool in_a = true;
  // Records: as few passes as 8-bit digits allow, but of EQUAL width (13 bits -> 7 + 6, not 8 + 5): the run a digit gets
  // inside a tile is tile / 2^width records, and the 4-byte key runs of an 8-bit pass (8 keys = 32 bytes) are partial
  // sectors. Keys-only sorts keep full 8-bit digits (fewest passes is what counts there).
  const int total_bits = end_bit - begin_bit;
  const int passes = (total_bits + 7) / 8;
  int width = records ? (total_bits + passes - 1) / passes : 8;
#ifdef PCV_EXPERIMENTS
  // PCV_SORT_FLIP=1 (libpcv_hip_exp.so): a two-pass record sort takes the NARROWER digit first (13 bits -> 6 + 7 instead of 7 + 6):
  // with PCV_REC_BLOCK=512 the first pass then runs two workgroups per CU on tiles of 4 096 with the runs of 64 records it has today
  static const bool flip = [] {
    const char* e = pcv_experiment("PCV_SORT_FLIP");
    return e && atoi(e) != 0;
  }();
  const bool flipped = flip && records && passes == 2 && (total_bits & 1) && map && rows;
  const int width2 = width;  // the second pass's width when flipped
  if (flipped) width = total_bits / 2;
#else
  constexpr bool flipped = false;
  const int width2 = width;
#endif
  for (int shift = begin_bit; shift < end_bit; shift += width) {
    int nbits = end_bit - shift < width ? end_bit - shift : width;
    uint32_t mask = (1u << nbits) - 1u;
    KeyT* src = in_a ? a : b;
    KeyT* dst = in_a ? b : a;
    // the histogram from the rank counts, the map applied inside the downsweep (12-byte records in tiles of 8 192; the map in
    // dynamic LDS next to the kernel's 107-117 KB: up to 16 384 half-word entries)
    const bool from_rows = map && rows && shift == begin_bit && rec12 && rec_variant == 3 && nbits <= 8;
    // half words in <= 32 KB beside the kernel's 107-117 KB; with a plane the kernel holds 137-146 KB: <= 20 / 10 KB of map (bigger maps are gathered)
    // (static LDS with a plane: 128 KB of tiles + 9.1 / 18.1 KB of digit state for 128 / 256 digit values)
    const bool map_in_lds = map_entries <= (with_plane ? (nbits <= 7 ? 10000u : 5000u) : 16384u);
    if (from_rows) {
      static const bool pass2_rows_on = [] {
        const char* e = pcv_experiment("PCV_SORT_ROWS2");  // 0 = the second pass counts its keys itself (experiments)
        return !e || atoi(e) != 0;
      }();
      const int nbits2 = flipped ? width2 : (end_bit - (shift + width) < width ? end_bit - (shift + width) : width);
      bool msd = false;
#ifdef PCV_EXPERIMENTS
      static const bool msd_on = [] {
        const char* e = pcv_experiment("PCV_SORT_MSD");  // 1: upper digit first, the second pass sorts inside every bucket
        return e && atoi(e) != 0;
      }();
      msd = msd_on;
#endif
      // (ranks of 15 bits — trees of up to 32 768 leaves — where the scratch holds their counters: rows_true_bins)
      const bool two = pass2_rows_on && map_entries <= rows_true_bins(n) && shift + width < end_bit && shift + width + nbits2 >= end_bit &&
                       (1u << total_bits) <= rows_true_bins(n) && nbits2 >= 1 && g.groups >= 8;
      msd = msd && two;
      // first / second pass: (shift, bits) of their digits — the lower digit first, unless msd
      const int p1_shift = msd ? shift + width : shift, p1_bits = msd ? nbits2 : nbits;
      const int p2_shift = msd ? shift : shift + width, p2_bits = msd ? nbits : nbits2;
      uint32_t* hist2 = totals + kRadix;
      uint32_t* totals2 = hist2 + (size_t)kRadix * kMaxGroups;
      uint2* ranges = reinterpret_cast<uint2*>(totals2 + kRadix);
      uint32_t* order = reinterpret_cast<uint32_t*>(ranges + kMaxGroups);
      uint32_t* rows_true = order + kMaxGroups;
      {
        PcvProf prof(ctx, PCV_K_SORT_HIST_ROWS);
        if (two) {  // (msd: the first pass takes the upper nbits2 bits, the second the lower nbits)
          static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&hist12_from_rows_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess;
          (void)ok;
          hipLaunchKernelGGL(hist12_from_rows_kernel, dim3(g.groups), dim3(256), std::min<size_t>((size_t)4 << (p1_bits + p2_bits), 131072), ctx->stream, rows,
                             map_entries, map, p1_bits, p2_bits, g.groups, hist, rows_true, msd ? 1 : 0);
        }
        else
          hipLaunchKernelGGL(hist_from_rows_kernel, dim3(g.groups), dim3(256), 0, ctx->stream, rows, map_entries, map, shift - begin_bit,
                             mask, g.groups, hist);
      }
      {
        PcvProf prof(ctx, PCV_K_SORT_SCAN);
        hipLaunchKernelGGL(scan_kernel, dim3(kRadix), dim3(256), 0, ctx->stream, hist, g.groups, totals);
      }
      // held-back second pass: its layout (two small kernels that need the histograms above, not the first pass) runs on the side
      // stream BESIDE the first pass; pcv_radix_sort_records_second joins the streams in front of the pass
      const bool side_layout = two && second && !msd && ctx->side && ctx->side_begin() == PCV_OK;
      const size_t dyn = map_in_lds ? (((size_t)map_entries * 2 + 15) & ~(size_t)15) : 0;
      const uint2* vin = (const uint2*)(in_a ? payload->vec_in : payload->vec_out);
      uint2* vout = (uint2*)(in_a ? payload->vec_out : payload->vec_in);
      {
        PcvProf prof(ctx, PCV_K_SORT_DOWNSWEEP_REC);
        // (the first pass of the sort: in_a is true here)
        const uint32_t* pin = with_plane ? (payload->first_in0 ? payload->first_in0 : in_a ? payload->in[0] : payload->out[0]) : nullptr;
        uint32_t* pout = with_plane ? (in_a ? payload->out[0] : payload->in[0]) : nullptr;
#define PCV_REC12_MAP(R, M, P)                                                                                                           \
  {                                                                                                                                      \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&downsweep_rec12_kernel<1024, 8, R, 4, false, M, P>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, P ? (R == 128 ? 20480 : 10240) : 32768) == hipSuccess; \
    (void)ok;                                                                                                                            \
    hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, R, 4, false, M, P>), dim3(g.groups), dim3(1024), dyn, ctx->stream,                \
                       (const uint32_t*)src, (uint32_t*)dst, n, g.chunk, g.groups, p1_shift, p1_bits, hist, totals, vin, vout, map,       \
                       map_entries, (const uint2*)nullptr, (const uint32_t*)nullptr, pin, pout, payload->color_in, payload->color_stride); \
  }
        // PCV_REC_WC (libpcv_hip_exp.so; bit 0: first pass, bit 1: second pass): the write-combining form of the downsweep
        // measured slower than the kernel that ships (profiles/r05_sort_same_box.json): not instantiated in libpcv_hip.so
        bool wc_done = false;
#ifdef PCV_EXPERIMENTS
        static const int rec_wc = [] {
          const char* e = pcv_experiment("PCV_REC_WC");
          return e ? atoi(e) : 0;
        }();
        if ((rec_wc & 1) && !with_plane && p1_bits <= 7 && !payload->color_in) {
          hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, 128, 4, false, 2, false, true>), dim3(g.groups), dim3(1024), 0, ctx->stream,
                             (const uint32_t*)src, (uint32_t*)dst, n, g.chunk, g.groups, p1_shift, p1_bits, hist, totals, vin, vout, map, map_entries,
                             (const uint2*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
          wc_done = true;
        }
#endif
#ifdef PCV_EXPERIMENTS
        static const int rec_block = [] {  // PCV_REC_BLOCK=512: the first pass in tiles of 4 096, two workgroups per CU
          const char* e = pcv_experiment("PCV_REC_BLOCK");
          return e ? atoi(e) : 1024;
        }();
        if (!wc_done && rec_block == 512 && !with_plane && p1_bits <= 7 && map_in_lds && dyn <= 28672 && !payload->color_in) {
          static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&downsweep_rec12_kernel<512, 8, 128, 4, false, 1, false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 28672) == hipSuccess;
          (void)ok;
          hipLaunchKernelGGL((downsweep_rec12_kernel<512, 8, 128, 4, false, 1, false>), dim3(g.groups), dim3(512), dyn, ctx->stream,
                             (const uint32_t*)src, (uint32_t*)dst, n, g.chunk, g.groups, p1_shift, p1_bits, hist, totals, vin, vout, map,
                             map_entries, (const uint2*)nullptr, (const uint32_t*)nullptr, pin, pout);
          wc_done = true;
        }
#endif
        if (wc_done) {
        } else if (with_plane) {
          if (p1_bits <= 7 && map_in_lds) PCV_REC12_MAP(128, 1, true)
          else if (map_in_lds) PCV_REC12_MAP(256, 1, true)
          else if (p1_bits <= 7) PCV_REC12_MAP(128, 2, true)
          else PCV_REC12_MAP(256, 2, true)
        } else {
          if (p1_bits <= 7 && map_in_lds) PCV_REC12_MAP(128, 1, false)
          else if (map_in_lds) PCV_REC12_MAP(256, 1, false)
          else if (p1_bits <= 7) PCV_REC12_MAP(128, 2, false)
          else PCV_REC12_MAP(256, 2, false)
        }
#undef PCV_REC12_MAP
      }
      in_a = !in_a;
      if (!two) continue;
      // second pass: pieces of whole first-pass runs, digit counts and ranges from rows_true and the first pass's offsets
      const int D1 = 1 << p1_bits;
      int blocks = kMaxGroups / D1;
      if (blocks > g.groups) blocks = g.groups;
      if (blocks < 1) blocks = 1;
      const int gpb = (g.groups + blocks - 1) / blocks;
      const int pieces = D1 * blocks;
      if (side_layout) std::swap(ctx->stream, ctx->side);
      {
        PcvProf prof(ctx, PCV_K_SORT_HIST_ROWS);
        hipLaunchKernelGGL(pass2_layout_kernel, dim3(pieces), dim3(256), 0, ctx->stream, rows_true, p1_bits, p2_bits, g.groups, blocks, gpb, hist,
                           totals, pieces, hist2, ranges, order);
      }
      {
        PcvProf prof(ctx, PCV_K_SORT_SCAN);
        if (msd)  // absolute positions of every (digit, piece) run inside its bucket; no digit prefix
          hipLaunchKernelGGL(msd_offsets_kernel, dim3(D1), dim3(256), 0, ctx->stream, hist2, pieces, blocks, p2_bits, totals, totals2);
        else
          hipLaunchKernelGGL(scan_kernel, dim3(kRadix), dim3(256), 0, ctx->stream, hist2, pieces, totals2);
      }
      if (side_layout) std::swap(ctx->stream, ctx->side);
      if (second && !msd) {  // the caller queues the pass itself (pcv_radix_sort_records_second)
        second->pending = true;
        second->join_side = side_layout;
        second->src = (const uint32_t*)(in_a ? a : b);
        second->dst = (uint32_t*)(in_a ? b : a);
        second->vec_src = in_a ? payload->vec_in : payload->vec_out;
        second->vec_dst = in_a ? payload->vec_out : payload->vec_in;
        second->plane_src = with_plane ? (in_a ? payload->in[0] : payload->out[0]) : nullptr;
        second->plane_dst = with_plane ? (in_a ? payload->out[0] : payload->in[0]) : nullptr;
        second->n = n, second->chunk = g.chunk;
        second->pieces = pieces, second->shift = p2_shift, second->nbits = p2_bits;
        second->low_bits = p1_bits, second->blocks = blocks;
        second->hist = hist2, second->totals = totals2, second->order = order, second->ranges = ranges;
        in_a = !in_a;
        break;
      }
      {
        PcvProf prof(ctx, PCV_K_SORT_DOWNSWEEP_REC);
        const uint32_t* src2 = (const uint32_t*)(in_a ? a : b);
        uint32_t* dst2 = (uint32_t*)(in_a ? b : a);
        const uint2* vin2 = (const uint2*)(in_a ? payload->vec_in : payload->vec_out);
        uint2* vout2 = (uint2*)(in_a ? payload->vec_out : payload->vec_in);
        const uint32_t* pin2 = with_plane ? (in_a ? payload->in[0] : payload->out[0]) : nullptr;
        uint32_t* pout2 = with_plane ? (in_a ? payload->out[0] : payload->in[0]) : nullptr;
#define PCV_REC12_P2(R, P)                                                                                                               \
  hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, R, 4, false, 0, P>), dim3(pieces), dim3(1024), 0, ctx->stream, src2, dst2, n, g.chunk, \
                     pieces, p2_shift, p2_bits, hist2, totals2, vin2, vout2, (const uint32_t*)nullptr, 0u, (const uint2*)ranges,            \
                     (const uint32_t*)order, pin2, pout2)
        bool wc2_done = false;
#ifdef PCV_EXPERIMENTS
        static const int rec_wc2 = [] {
          const char* e = pcv_experiment("PCV_REC_WC");
          return e ? atoi(e) : 0;
        }();
        if ((rec_wc2 & 2) && !with_plane && p2_bits <= 7) {
          hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, 128, 4, false, 0, false, true>), dim3(pieces), dim3(1024), 0, ctx->stream, src2, dst2, n,
                             g.chunk, pieces, p2_shift, p2_bits, hist2, totals2, vin2, vout2, (const uint32_t*)nullptr, 0u, (const uint2*)ranges,
                             (const uint32_t*)order, (const uint32_t*)nullptr, (uint32_t*)nullptr);
          wc2_done = true;
        }
#endif
        if (wc2_done) {
        } else if (with_plane) {
          if (p2_bits <= 7) PCV_REC12_P2(128, true);
          else PCV_REC12_P2(256, true);
        } else {
          if (p2_bits <= 7) PCV_REC12_P2(128, false);
          else PCV_REC12_P2(256, false);
        }
#undef PCV_REC12_P2
      }
      in_a = !in_a;
      break;
    }
    if (map && shift == begin_bit && sizeof(KeyT) == 4 && payload && payload->vec_in) {
      PcvProf prof(ctx, PCV_K_SORT_UPSWEEP_MAP);  // finalize fused into the first upsweep
      const bool lds = map_entries && map_entries <= kMapLdsEntries;
      const size_t dyn = lds ? (size_t)map_entries * 4 : 0;
      void* pay = in_a ? payload->vec_in : payload->vec_out;
#define PCV_UPSWEEP_MAP(L, C)                                                                                                    \
  hipLaunchKernelGGL((upsweep_map_kernel<L, C>), dim3(g.groups), dim3(kBlock), dyn, ctx->stream, (uint32_t*)src, n, g.chunk, g.groups, \
                     shift, mask, hist, map, map_entries, pay)
      if (lds && compact) PCV_UPSWEEP_MAP(true, true);
      else if (lds) PCV_UPSWEEP_MAP(true, false);
      else if (compact) PCV_UPSWEEP_MAP(false, true);
      else PCV_UPSWEEP_MAP(false, false);
#undef PCV_UPSWEEP_MAP
    } else {
      PcvProf prof(ctx, sizeof(KeyT) == 8 ? PCV_K_SORT_UPSWEEP64 : PCV_K_SORT_UPSWEEP32);
      static const bool plain_on = [] {
        const char* e = pcv_experiment("PCV_UPSWEEP_PLAIN");  // 0 = the ballot match everywhere (experiments)
        return !e || atoi(e) != 0;
      }();
      if (plain_on && records && shift != begin_bit)  // the upper digits of a record sort, after a pass has mixed them
        hipLaunchKernelGGL((upsweep_kernel<KeyT, true>), dim3(g.groups), dim3(kBlock), 0, ctx->stream, src, n, g.chunk, g.groups, shift,
                           mask, hist);
      else
        hipLaunchKernelGGL((upsweep_kernel<KeyT, false>), dim3(g.groups), dim3(kBlock), 0, ctx->stream, src, n, g.chunk, g.groups, shift,
                           mask, hist);
    }
    {
      PcvProf prof(ctx, PCV_K_SORT_SCAN);
      hipLaunchKernelGGL(scan_kernel, dim3(kRadix), dim3(256), 0, ctx->stream, hist, g.groups, totals);
    }
    if (!records) {
      PcvProf prof(ctx, sizeof(KeyT) == 8 ? PCV_K_SORT_DOWNSWEEP64 : PCV_K_SORT_DOWNSWEEP32);
      hipLaunchKernelGGL(downsweep_keys_kernel<KeyT>, dim3(g.groups), dim3(kBlock), 0, ctx->stream, src, dst, n, g.chunk,
                         g.groups, shift, nbits, hist, totals);
    } else {
      RecPtrs rp{};
      rp.vec_in = in_a ? payload->vec_in : payload->vec_out;
      rp.vec_out = in_a ? payload->vec_out : payload->vec_in;
      rp.nplanes = payload->nwords;
      for (int w = 0; w < payload->nwords; ++w) {
        rp.plane_in[w] = in_a ? payload->in[w] : payload->out[w];
        rp.plane_out[w] = in_a ? payload->out[w] : payload->in[w];
      }
      if (shift == begin_bit && payload->nwords > 0 && payload->first_in0) rp.plane_in[0] = payload->first_in0;
      static const bool prefetch = [] {
        const char* e = pcv_experiment("PCV_REC_PREFETCH");  // 0 = the unpipelined kernel (experiments)
        return !e || atoi(e) != 0;
      }();
      PcvProf prof(ctx, PCV_K_SORT_DOWNSWEEP_REC);
      if (rec12)
        rec12_launch(ctx, rec_variant, g, (const uint32_t*)src, (uint32_t*)dst, n, shift, nbits, hist, totals, (const uint2*)rp.vec_in,
                     (uint2*)rp.vec_out, with_plane ? rp.plane_in[0] : nullptr, with_plane ? rp.plane_out[0] : nullptr);
      else if (compact)
        hipLaunchKernelGGL((downsweep_rec_kernel<true, true, uint2>), dim3(g.groups), dim3(kBlock), 0, ctx->stream, (const uint32_t*)src,
                           (uint32_t*)dst, n, g.chunk, g.groups, shift, nbits, hist, totals, rp);
      else if (payload->vec_in && prefetch)
        hipLaunchKernelGGL((downsweep_rec_kernel<true, true>), dim3(g.groups), dim3(kBlock), 0, ctx->stream, (const uint32_t*)src,
                           (uint32_t*)dst, n, g.chunk, g.groups, shift, nbits, hist, totals, rp);
      else if (payload->vec_in)
        hipLaunchKernelGGL(downsweep_rec_kernel<true>, dim3(g.groups), dim3(kBlock), 0, ctx->stream, (const uint32_t*)src,
                           (uint32_t*)dst, n, g.chunk, g.groups, shift, nbits, hist, totals, rp);
      else
        hipLaunchKernelGGL(downsweep_rec_kernel<false>, dim3(g.groups), dim3(kBlock), 0, ctx->stream, (const uint32_t*)src,
                           (uint32_t*)dst, n, g.chunk, g.groups, shift, nbits, hist, totals, rp);
    }
    in_a = !in_a;
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  *result_in_a = in_a;
  return PCV_OK;
}


// ---- the sample's key sort with ONE launch per digit ("onesweep"): measured, no faster, libpcv_hip_exp.so only --------------
// Round 6: upsweep + scan + downsweep per 8-bit digit are fifteen dependent launches of 6-14 us for the 1.5 M sample keys of a
// 100 M-point build. One kernel per 9-bit digit (tiles of 8 192 keys, digit counts published per tile, a look-back over the tiles
// before) is four launches + one counting pass — and takes the same time (tools/key_sort_probe.py, 1.56 M keys of 36 bits:
// 110 us either way; 15.6 M keys: 495 against 610): the kernel without its look-back runs 14 us per pass, the look-back costs
// another 11 — what one workgroup publishes reaches another XCD's workgroup through memory, not through its L2 (an agent-scope
// release fence writes the whole L2 back: 25 us), so two dependent hand-overs inside the kernel cost what the two extra launches
// did. Kept for the record (PCV_SAMPLE_ONESWEEP=1); profiles/r06_ab_sample_key_sort_one_launch_per_digit_dropped.json.
#ifdef PCV_EXPERIMENTS
// The sample of the single-chain build is 1.5 M keys (12.5 MB) at 100 M points: upsweep + scan + downsweep per 8-bit digit
// were fifteen dependent launches of 6-14 us each, every one of them bound by its own launch and drain. Here the digit counts
// of ALL passes are taken once (key_hist_kernel), and every pass is
// ONE kernel over tiles of 8 192 keys: a tile ranks its keys in LDS, publishes its digit counts (512 values + one flag word), adds
// up the counts of the tiles before it — back to the nearest tile whose inclusive prefix is already there; with all 191 tiles of
// a 100 M-point build's sample resident at once that is usually tile 0, and the sum is 190 independent loads per lane instead of
// a chain of dependent ones (a look-back over value+flag words, 32 at a time, cost 11 us per pass; this one ~3) — and writes its runs.
// Tiles are handed out by a ticket, so a tile only ever waits for tiles that are already running. Digits are up to 9 bits wide
// (512 lanes, lane t owns digit t): 36 bits of key are four launches.
constexpr int kOneBlock = 512, kOneKpt = 16, kOneTile = kOneBlock * kOneKpt, kOneWaves = kOneBlock / 64, kOneRadix = 512;
constexpr int kOneHistGroups = 128;   // workgroups of key_hist_kernel
constexpr int kOneMaxPasses = 8;

struct OnePasses {
  int passes;
  int shift[kOneMaxPasses], bits[kOneMaxPasses];
};

__global__ __launch_bounds__(kOneBlock) void key_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, OnePasses ps,
                                                             uint32_t* __restrict__ ghist /* [passes][512], zero */) {
  __shared__ uint32_t h[kOneMaxPasses][kOneRadix];
  for (int i = threadIdx.x; i < ps.passes * kOneRadix; i += kOneBlock) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint32_t per = ((n + kOneHistGroups - 1) / kOneHistGroups + 1u) & ~1u;  // even: 16-byte loads stay aligned
  const uint32_t begin = min(n, blockIdx.x * per), end = min(n, begin + per);
  typedef uint64_t Vec2 __attribute__((ext_vector_type(2)));
  uint32_t i = begin + 2u * threadIdx.x;
  for (; i + 2u * kOneBlock * 3u + 2u <= end; i += 2u * kOneBlock * 4u) {  // four loads in flight
    Vec2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const Vec2*>(keys + i + 2u * kOneBlock * u);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        for (int p = 0; p < ps.passes; ++p) atomicAdd(&h[p][(uint32_t)(v[u][k] >> ps.shift[p]) & ((1u << ps.bits[p]) - 1u)], 1u);
  }
  for (; i < end; i += 2u * kOneBlock)
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (i + k < end) {
        const uint64_t key = keys[i + k];
        for (int p = 0; p < ps.passes; ++p) atomicAdd(&h[p][(uint32_t)(key >> ps.shift[p]) & ((1u << ps.bits[p]) - 1u)], 1u);
      }
  __syncthreads();
  for (int j = threadIdx.x; j < ps.passes * kOneRadix; j += kOneBlock) {
    const uint32_t c = (&h[0][0])[j];
    if (c) atomicAdd(&ghist[j], c);  // 128 workgroups x (a few hundred non-empty digits): ghist was cleared with the tickets
  }
}

// exclusive prefix over the 512 lanes of the workgroup (lane t -> sum of v of lanes < t); `tot` (8 words of LDS) is scratch
__device__ __forceinline__ uint32_t one_block_exclusive(uint32_t v, uint32_t* tot, int lane, int wave) {
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) tot[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int w = 0; w < kOneWaves; ++w) woff += w < wave ? tot[w] : 0u;
  __syncthreads();
  return woff + inc - v;
}

// keeps in (plo, phi) the lanes whose digit agrees with this lane's in bit B
template <int B>
__device__ __forceinline__ void one_match_bit(uint32_t d, uint32_t& plo, uint32_t& phi) {
  int m;  // all ones when bit B of the digit is set
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(d), "n"(B));
  const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
  plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, (uint32_t)m, 0x90);  // p & ~(ballot ^ m)
  phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
}

__global__ __launch_bounds__(kOneBlock) void onesweep_keys_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n,
                                                                  int shift, int nbits, const uint32_t* __restrict__ ghist /* [512] of this pass */,
                                                                  uint32_t* __restrict__ ticket, uint32_t* __restrict__ flags /* [tiles], zero */,
                                                                  uint32_t* __restrict__ vals /* [tiles][2][512] */, int diag) {
  extern __shared__ uint64_t skeys[];  // kOneTile keys (64 KB)
  __shared__ uint32_t whist[kOneWaves][kOneRadix];
  __shared__ uint32_t delta[kOneRadix];
  __shared__ uint32_t tot[kOneWaves];
  __shared__ uint32_t s_tile;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
  for (int w = 0; w < kOneWaves; ++w) whist[w][t] = 0;
  const uint32_t gcount = ghist[t];  // the digit's count over the whole input (key_hist_kernel)
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t base = tile * (uint32_t)kOneTile;
  const uint32_t tile_n = min((uint32_t)kOneTile, n - base);
  const uint32_t mask = (1u << nbits) - 1u;
  const uint32_t wbase = wave * 64 * kOneKpt + lane;
  uint64_t key[kOneKpt];
#pragma unroll
  for (int i = 0; i < kOneKpt; ++i) {
    const uint32_t li = wbase + i * 64;
    key[i] = li < tile_n ? in[base + li] : 0ull;
    if (diag & 8) key[i] = (uint64_t)li << shift;  // timing only (pcv_exp_time_key_sort)
  }
  // rank of every key among the earlier keys of its digit inside the wave's slice (stable: iteration-major, lane-minor)
  uint16_t lpos[kOneKpt];
#pragma unroll
  for (int i = 0; i < kOneKpt; ++i) {
    const bool valid = wbase + i * 64 < tile_n;
    const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
    if (diag & 2) {
      lpos[i] = 0;
      continue;
    }
    const uint64_t vm = __ballot(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
    one_match_bit<0>(d, plo, phi), one_match_bit<1>(d, plo, phi), one_match_bit<2>(d, plo, phi), one_match_bit<3>(d, plo, phi);
    one_match_bit<4>(d, plo, phi);  // (bits above the digit's width are zero in every lane: matching them changes nothing)
    if (nbits > 5) one_match_bit<5>(d, plo, phi);  // wave-uniform
    if (nbits > 6) one_match_bit<6>(d, plo, phi);
    if (nbits > 7) one_match_bit<7>(d, plo, phi);
    if (nbits > 8) one_match_bit<8>(d, plo, phi);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    uint32_t* slot = &whist[wave][d];
    const uint32_t pre = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (valid && below == 0)
      (void)__hip_atomic_fetch_add(slot, (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lpos[i] = (uint16_t)(pre + below);
  }
  __syncthreads();
  uint32_t pre_w[kOneWaves], mine = 0;
#pragma unroll
  for (int w = 0; w < kOneWaves; ++w) {
    pre_w[w] = mine;
    mine += whist[w][t];
  }
  // This tile's digit counts for the tiles after it: the 512 values first, then ONE flag word (1 = counts there, 2 = inclusive
  // prefix there too). Polling touches the flags only; the values are summed with independent loads once they are known to be there.
  uint32_t* agg = vals + (size_t)tile * 2u * kOneRadix;  // [tile][0] counts, [tile][1] inclusive prefix
  __hip_atomic_store(agg + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tile == 0) __hip_atomic_store(agg + kOneRadix + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // No fence: an agent-scope release writes back the whole L2 of the XCD (25 us here). The values are agent-scope atomic stores
  // (written through on their own); once they are acknowledged (vmcnt 0) in every lane, the flag may follow.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) __hip_atomic_store(flags + tile, tile == 0 ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t excl = 0;
  if (tile > 0 && !(diag & 1)) {
    if (wave == 0) {  // lane l looks at tile hi - l: the nearest tile with a prefix, once every tile after it has its counts
      int64_t hi = (int64_t)tile - 1, found = -1;
      uint32_t spins = 0;
      while (found < 0) {
        const int64_t j = hi - lane;
        const uint32_t f = j >= 0 ? __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
        const uint64_t pm = __ballot(f == 2u);
        const int first = pm ? (int)__builtin_ctzll(pm) : 64;
        if (__ballot(f == 0u && lane < first)) {  // counts missing in front of it: look again
          if (++spins > (1u << 24)) __builtin_trap();  // a tile that never publishes: fail loudly, never hang
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        if (first < 64) found = hi - first;
        else hi -= 64;
      }
      if (lane == 0) s_tile = (uint32_t)found;
    }
    __syncthreads();
    const uint32_t from = s_tile;
    excl = __hip_atomic_load(vals + ((size_t)from * 2u + 1u) * kOneRadix + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t j = from + 1u;
    for (; j + 16u <= tile; j += 16u) {
      uint32_t v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = __hip_atomic_load(vals + (size_t)(j + u) * 2u * kOneRadix + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 16; ++u) excl += v[u];
    }
    for (; j < tile; ++j) excl += __hip_atomic_load(vals + (size_t)j * 2u * kOneRadix + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(agg + kOneRadix + t, excl + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(flags + tile, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const uint32_t gstart = one_block_exclusive(gcount, tot, lane, wave);  // keys of smaller digits in the whole input
  const uint32_t lstart = one_block_exclusive(mine, tot, lane, wave);    // ... in this tile
#pragma unroll
  for (int w = 0; w < kOneWaves; ++w) whist[w][t] = lstart + pre_w[w];
  delta[t] = gstart + excl - lstart;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kOneKpt; ++i)
    if (wbase + i * 64 < tile_n) skeys[whist[wave][(uint32_t)(key[i] >> shift) & mask] + lpos[i]] = key[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kOneKpt; ++j) {
    const uint32_t p = j * kOneBlock + t;
    if (p < tile_n) {
      const uint64_t k = skeys[diag & 2 ? p : p];
      uint32_t at = delta[(uint32_t)(k >> shift) & mask] + p;
      if (diag) at = (diag & 4) ? p : min(at, n - 1u);  // timing-only variants: anywhere inside the buffer
      out[at] = k;
    }
  }
}

#endif  // PCV_EXPERIMENTS (onesweep kernels)

}  // namespace

#ifdef PCV_EXPERIMENTS
// The sample's key sort (see onesweep_keys_kernel). scratch: [tickets 64 words | digit counts passes x 512 | flags passes x tiles |
// values passes x tiles x 2 x 512]; everything in front of the values ZERO when key_hist_kernel starts — pcv_onesweep_zero_words(n,
// bits) words (the caller lets the kernel that writes the keys clear them: no launch of its own).
static int onesweep_passes(int bits) { return (bits + 8) / 9; }
bool pcv_onesweep_fits(uint64_t n, int bits) { return n > 0 && n < (1ull << 30) && bits > 0 && onesweep_passes(bits) <= kOneMaxPasses; }
size_t pcv_onesweep_zero_words(uint64_t n, int bits) {
  const size_t tiles = (size_t)((n + kOneTile - 1) / kOneTile);
  return 64 + (size_t)onesweep_passes(bits) * (kOneRadix + ((tiles + 3) & ~(size_t)3));
}
size_t pcv_onesweep_scratch_words(uint64_t n, int bits) {
  const size_t tiles = (size_t)((n + kOneTile - 1) / kOneTile);
  return ((pcv_onesweep_zero_words(n, bits) + 63) & ~(size_t)63) + (size_t)onesweep_passes(bits) * tiles * 2 * kOneRadix;
}
int pcv_sort_keys_onesweep(pcv_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, int begin_bit, int end_bit, uint32_t* scratch,
                           bool* result_in_a, int diag) {
  const int bits = end_bit - begin_bit;
  if (!pcv_onesweep_fits(n, bits)) return ctx->fail(PCV_E_INVALID, "onesweep key sort: size");
  OnePasses ps{};
  ps.passes = onesweep_passes(bits);
  for (int p = 0, at = begin_bit; p < ps.passes; ++p) {  // digits as even as they come: 36 bits = 4 x 9, 39 = 5 x 8 (the last one 7)
    const int w = (end_bit - at + (ps.passes - p) - 1) / (ps.passes - p);
    ps.shift[p] = at, ps.bits[p] = w, at += w;
  }
  const uint32_t tiles = (uint32_t)((n + kOneTile - 1) / kOneTile);
  uint32_t* tickets = scratch;
  const size_t tiles4 = ((size_t)tiles + 3) & ~(size_t)3;
  uint32_t* ghist = scratch + 64;
  uint32_t* flags = ghist + (size_t)ps.passes * kOneRadix;
  uint32_t* vals = scratch + ((pcv_onesweep_zero_words(n, bits) + 63) & ~(size_t)63);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&onesweep_keys_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  kOneTile * 8) == hipSuccess;
  if (!attr_ok) return ctx->fail(PCV_E_HIP, "onesweep key sort: LDS");
  {
    PcvProf prof(ctx, PCV_K_SORT_UPSWEEP64);
    hipLaunchKernelGGL(key_hist_kernel, dim3(kOneHistGroups), dim3(kOneBlock), 0, ctx->stream, keys_a, (uint32_t)n, ps, ghist);
  }
  bool in_a = true;
  for (int p = 0; p < ps.passes; ++p) {
    PcvProf prof(ctx, PCV_K_SORT_DOWNSWEEP64);
    hipLaunchKernelGGL(onesweep_keys_kernel, dim3(tiles), dim3(kOneBlock), kOneTile * 8, ctx->stream, in_a ? keys_a : keys_b, in_a ? keys_b : keys_a,
                       (uint32_t)n, ps.shift[p], ps.bits[p], ghist + (size_t)p * kOneRadix, tickets + p,
                       flags + (size_t)p * tiles4, vals + (size_t)p * tiles * 2 * kOneRadix, diag);
    in_a = !in_a;
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  *result_in_a = in_a;
  return PCV_OK;
}
#endif  // PCV_EXPERIMENTS (onesweep host side)

// two histograms + totals, the second pass's piece ranges, and the rank counts re-indexed by true rank (16 384 per sort
// workgroup, 64 MB) — the last only for inputs whose record sort can take the two-pass rows path at all (12-byte records in
// tiles of 8 192, i.e. >= 8 sort workgroups: n >= 65 536); small builds (tests, virtual ranks) get by with 2 MB
size_t pcv_sort_scratch_bytes(uint64_t n) {
  const size_t rows_true = make_geom(n, 8192).groups >= 8 ? rows_true_bins(n) * (size_t)kMaxGroups : 0;
  return (2 * ((size_t)kRadix * kMaxGroups + kRadix) + 3 * (size_t)kMaxGroups + rows_true) * sizeof(uint32_t) + 256;
}

// Does the first pass of a mapped 12-byte record sort with rank-count rows read the records' colour itself (PcvSortPayload::color_in)?
// It does in the form that ships (the rows-based downsweep_rec12_kernel<..., MAP != 0>); the experiment variants of the record
// kernel (PCV_REC_VARIANT, libpcv_hip_exp.so) do not.
bool pcv_sort_first_pass_joins_color(uint64_t n) {
  static const int rec_variant = [] {
    const char* e = pcv_experiment("PCV_REC_VARIANT");
    return e ? atoi(e) : 3;
  }();
  return n > 0 && rec_variant == 3;
}

int pcv_radix_sort_u64(pcv_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, int begin_bit, int end_bit,
                       PcvSortPayload* payload, void* scratch, bool* result_in_a) {
  return radix_sort<uint64_t>(ctx, keys_a, keys_b, n, begin_bit, end_bit, payload, scratch, result_in_a);
}
int pcv_radix_sort_u32(pcv_ctx* ctx, uint32_t* keys_a, uint32_t* keys_b, uint64_t n, int begin_bit, int end_bit,
                       PcvSortPayload* payload, void* scratch, bool* result_in_a) {
  return radix_sort<uint32_t>(ctx, keys_a, keys_b, n, begin_bit, end_bit, payload, scratch, result_in_a);
}
// Record sort whose first upsweep also translates the ranks through `map` (single-chain build); 12-byte records
// (payload->vec_bytes == 8): the rank sits in bits 8.. of the key
int pcv_radix_sort_records_mapped(pcv_ctx* ctx, uint32_t* keys_a, uint32_t* keys_b, uint64_t n, int key_bits,
                                  PcvSortPayload* payload, void* scratch, const uint32_t* map, uint32_t map_entries,
                                  bool* result_in_a, const uint32_t* rows, PcvSortSecond* second) {
  const int base = payload && payload->vec_bytes == 8 ? 8 : 0;
  return radix_sort<uint32_t>(ctx, keys_a, keys_b, n, base, base + key_bits, payload, scratch, result_in_a, map, map_entries, rows, second);
}
int pcv_radix_sort_records_second(pcv_ctx* ctx, PcvSortSecond* sd, const PcvSortFuse* fuse) {
  if (!sd || !sd->pending) return PCV_OK;
  sd->pending = false;
  if (sd->join_side && ctx->side_end() != PCV_OK) return ctx->fail(PCV_E_HIP, "record sort: side stream");
#define PCV_REC12_SECOND(R, P)                                                                                                           \
  hipLaunchKernelGGL((downsweep_rec12_kernel<1024, 8, R, 4, false, 0, P, false>), dim3(sd->pieces), dim3(1024), 0, ctx->stream, sd->src,     \
                     sd->dst, sd->n, sd->chunk, sd->pieces, sd->shift, sd->nbits, sd->hist, sd->totals, (const uint2*)sd->vec_src,         \
                     (uint2*)sd->vec_dst, (const uint32_t*)nullptr, 0u, (const uint2*)sd->ranges, sd->order, sd->plane_src, sd->plane_dst)
#define PCV_REC12_SETTLE(P, ARG)                                                                                                         \
  hipLaunchKernelGGL((downsweep_settle_kernel<P>), dim3(sd->pieces), dim3(1024), 0, ctx->stream, sd->src, sd->dst, sd->n, sd->chunk,      \
                     sd->pieces, sd->shift, sd->nbits, sd->hist, sd->totals, (const uint2*)sd->vec_src, (uint2*)sd->vec_dst,              \
                     (const uint2*)sd->ranges, sd->order, sd->plane_src, sd->plane_dst, ARG)
  const bool plane = sd->plane_src != nullptr;
  if (fuse && (sd->nbits <= 7 || !plane) && (!plane || fuse->inten_blob)) {
    PcvProf prof(ctx, PCV_K_SORT_SETTLE);
    PcvSortFuse fz = *fuse;
    fz.low_bits = (uint32_t)sd->low_bits, fz.blocks = (uint32_t)sd->blocks;
    // colour-only records: tiles of 4 096 (512 lanes), TWO workgroups per CU — the pass is bound by its own phases (loads, LDS
    // ranking, barriers), not by bytes, and a second workgroup fills them: 0.69-0.70 -> 0.63-0.64 ms at 100 M points in one call
    // (tiles of 2 048, four workgroups: 0.75). With the intensity plane two workgroups' LDS does not fit: tiles of 8 192.
    int settle_block = 512;
#ifdef PCV_EXPERIMENTS
    static const int settle_block_env = [] {  // PCV_SETTLE_BLOCK=1024 / 256 (libpcv_hip_exp.so)
      const char* e = pcv_experiment("PCV_SETTLE_BLOCK");
      return e ? atoi(e) : 0;
    }();
    if (settle_block_env) settle_block = settle_block_env;
#endif
#define PCV_REC12_SETTLE_B(B)                                                                                                            \
  hipLaunchKernelGGL((downsweep_settle_kernel<false, B>), dim3(sd->pieces), dim3(B), 0, ctx->stream, sd->src, sd->dst, sd->n, sd->chunk,  \
                     sd->pieces, sd->shift, sd->nbits, sd->hist, sd->totals, (const uint2*)sd->vec_src, (uint2*)sd->vec_dst,              \
                     (const uint2*)sd->ranges, sd->order, sd->plane_src, sd->plane_dst, fz)
    if (sd->nbits > 7)  // ranks of 16 bits, colour-only: 256 digit values, tiles of 8 192
      hipLaunchKernelGGL((downsweep_settle_kernel<false, 1024, 256>), dim3(sd->pieces), dim3(1024), 0, ctx->stream, sd->src, sd->dst, sd->n,
                         sd->chunk, sd->pieces, sd->shift, sd->nbits, sd->hist, sd->totals, (const uint2*)sd->vec_src, (uint2*)sd->vec_dst,
                         (const uint2*)sd->ranges, sd->order, sd->plane_src, sd->plane_dst, fz);
#ifdef PCV_EXPERIMENTS
    else if (plane && settle_block == 512 && settle_block_env == 512)  // PCV_SETTLE_BLOCK=512 with the plane: two workgroups fill the LDS exactly
      hipLaunchKernelGGL((downsweep_settle_kernel<true, 512>), dim3(sd->pieces), dim3(512), 0, ctx->stream, sd->src, sd->dst, sd->n, sd->chunk,
                         sd->pieces, sd->shift, sd->nbits, sd->hist, sd->totals, (const uint2*)sd->vec_src, (uint2*)sd->vec_dst,
                         (const uint2*)sd->ranges, sd->order, sd->plane_src, sd->plane_dst, fz);
#endif
    else if (plane) PCV_REC12_SETTLE(true, fz);
#ifdef PCV_EXPERIMENTS
    else if (settle_block == 1024) PCV_REC12_SETTLE_B(1024);
    else if (settle_block == 256) PCV_REC12_SETTLE_B(256);
#endif
    else PCV_REC12_SETTLE_B(512);
#undef PCV_REC12_SETTLE_B
    (void)settle_block;
  } else {
    // (the caller decides with the same condition whether to pass `fuse`, pcv_build_finish `fuse_sort`; should the two ever drift
    // apart the build must not degrade silently into leaves nobody settles: one message per cause)
    if (fuse && plane && sd->nbits > 7)
      return ctx->fail(PCV_E_INVALID, "record sort: the settling pass with an intensity plane needs a second digit of <= 7 bits");
    if (fuse) return ctx->fail(PCV_E_INVALID, "record sort: the settling pass with an intensity plane needs the octree's intensity blob");
    PcvProf prof(ctx, PCV_K_SORT_DOWNSWEEP_REC);
    if (sd->nbits <= 7 && plane) PCV_REC12_SECOND(128, true);
    else if (sd->nbits <= 7) PCV_REC12_SECOND(128, false);
    else if (plane) PCV_REC12_SECOND(256, true);
    else PCV_REC12_SECOND(256, false);
  }
#undef PCV_REC12_SETTLE
#undef PCV_REC12_SECOND
  return hipGetLastError() == hipSuccess ? PCV_OK : ctx->fail(PCV_E_HIP, "record sort: second pass");
}
void pcv_sort_rec12_geometry(uint64_t n, int* groups, uint64_t* chunk) {
  const SortGeom g = make_geom(n, 8192);
  *groups = g.groups;
  *chunk = g.chunk;
}

#ifdef PCV_EXPERIMENTS
// libpcv_hip_exp.so only: times `iters` key sorts of n pseudo-random `bits`-bit keys (HIP events around each sort);
// onesweep != 0: pcv_sort_keys_onesweep (diag = timing-only variants of its kernel), else the three-kernel radix sort.
__global__ void exp_fill_keys_kernel(uint64_t* k, uint32_t n, int bits, int top) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t)i * 0x9e3779b97f4a7c15ull + 0x1234567ull;
  x ^= x >> 29, x *= 0xbf58476d1ce4e5b9ull, x ^= x >> 32;
  k[i] = (x & ((1ull << bits) - 1ull)) << (top - bits);
}
extern "C" int pcv_exp_time_key_sort(pcv_ctx* ctx, uint64_t n, int bits, int onesweep, int diag, int iters, float* ms_out) {
  void *a = nullptr, *b = nullptr, *sc = nullptr, *sc2 = nullptr;
  const int top = 3 * PCV_MAX_KEY_LEVELS;
  int rc;
  if ((rc = ctx->dev_alloc(&a, n * 8 + 256)) || (rc = ctx->dev_alloc(&b, n * 8 + 256)) ||
      (rc = ctx->dev_alloc(&sc, pcv_onesweep_scratch_words(n, bits) * 4 + 64)) || (rc = ctx->dev_alloc(&sc2, pcv_sort_scratch_bytes(n))))
    return rc;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(exp_fill_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint64_t*)a, (uint32_t)n, bits, top);
    (void)hipMemsetAsync(sc, 0, pcv_onesweep_zero_words(n, bits) * 4, ctx->stream);
    (void)hipEventRecord(e0, ctx->stream);
    bool in_a;
    if (onesweep) rc = pcv_sort_keys_onesweep(ctx, (uint64_t*)a, (uint64_t*)b, n, top - bits, top, (uint32_t*)sc, &in_a, diag);
    else rc = pcv_radix_sort_u64(ctx, (uint64_t*)a, (uint64_t*)b, n, top - bits, top, nullptr, sc2, &in_a);
    (void)hipEventRecord(e1, ctx->stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms_out[it], e0, e1);
    if (rc) break;
  }
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  ctx->dev_free(a), ctx->dev_free(b), ctx->dev_free(sc), ctx->dev_free(sc2);
  return rc;
}
#endif
