// pcv_sort.hip — K3: stable LSD radix sort for gfx950 (wave64), 8-bit digits, LDS histograms.
//
// The reference never sorts: it partitions every node's stream into 8 child files level by level
// (src/octree/generation.rs:58-126: 8 clones + 8 `retain`s per batch). On the GPU the same *stable* grouping
// is one radix sort of the path keys (and later of leaf-rank records), SURVEY.md §8a R7 / F11.
//
// Structure per 8-bit pass (reduce-then-scan, no inter-workgroup spinning):
//   upsweep   : G workgroups, each counts the digits of its contiguous chunk in per-wave LDS histograms
//   scan      : one workgroup turns the 256 x G counts (digit-major) into exclusive global offsets
//   downsweep : the same G workgroups walk their chunk tile by tile; inside a tile each wave ranks its keys
//               with ballot-built peer masks (64-lane match-any), a 256-entry LDS scan orders the digits,
//               keys (and any payload words) are staged through LDS so global stores go out as runs.
// HBM traffic per pass and key: sizeof(key) (upsweep) + 2*sizeof(key) + 8 B per payload word.
#include "pcv_internal.h"

namespace {

constexpr int kBlock = 256;  // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kKpt = 16;  // keys per lane per tile
constexpr int kTile = kBlock * kKpt;
constexpr int kRadix = 256;
constexpr int kMaxGroups = 1024;

struct SortGeom {
  uint64_t n;
  uint64_t chunk;  // keys per workgroup, multiple of kTile
  int groups;
};

SortGeom make_geom(uint64_t n) {
  SortGeom g;
  g.n = n;
  uint64_t tiles = (n + kTile - 1) / kTile;
  uint64_t tiles_per_group = (tiles + kMaxGroups - 1) / kMaxGroups;
  if (tiles_per_group == 0) tiles_per_group = 1;
  g.chunk = tiles_per_group * kTile;
  g.groups = (int)((n + g.chunk - 1) / g.chunk);
  if (g.groups < 1) g.groups = 1;
  return g;
}

template <typename KeyT>
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
  // chunk is a multiple of kTile and buffers come from the pool (256-B aligned) => 16-byte loads are aligned
  uint64_t i = begin + (uint64_t)threadIdx.x * kVec;
  for (; i + kVec <= end; i += (uint64_t)kBlock * kVec) {
    VecT v = *reinterpret_cast<const VecT*>(keys + i);
#pragma unroll
    for (int k = 0; k < kVec; ++k) atomicAdd(&wh[wave][(uint32_t)(v[k] >> shift) & mask], 1u);
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

// One workgroup per digit: exclusive scan of that digit's `groups` counters in place (groups <= 1024) and the
// digit's total. The scan across digits is folded into the downsweep prologue (256 values).
__global__ __launch_bounds__(256) void scan_kernel(uint32_t* __restrict__ hist, int groups,
                                                    uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_tot[4];
  uint32_t* row = hist + (uint64_t)blockIdx.x * groups;
  const int per = (groups + 255) / 256;  // <= 4
  const int begin = threadIdx.x * per;
  uint32_t v[4] = {0, 0, 0, 0};
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
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
  for (int i = 0; i < 4; ++i)
    if (i < per && begin + i < groups) {
      row[begin + i] = run;
      run += v[i];
    }
  if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

struct PayloadPtrs {
  int nwords;
  const uint32_t* in[8];
  uint32_t* out[8];
};

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void downsweep_kernel(const KeyT* __restrict__ keys_in,
                                                            KeyT* __restrict__ keys_out, uint64_t n, uint64_t chunk,
                                                            int groups, int shift, int nbits,
                                                            const uint32_t* __restrict__ offsets /* [256][groups] */,
                                                            const uint32_t* __restrict__ totals /* [256] */,
                                                            PayloadPtrs pl) {
  __shared__ KeyT skeys[kTile];
  __shared__ uint32_t whist[kWaves][kRadix];
  __shared__ uint32_t digit_base[kRadix];
  __shared__ uint32_t tile_start[kRadix];
  __shared__ uint32_t tile_count[kRadix];
  __shared__ uint32_t wave_tot[kWaves];
  uint32_t* stage32 = reinterpret_cast<uint32_t*>(skeys);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  {
    // global base of digit t for this workgroup = (keys with a smaller digit) + (same digit, earlier workgroups)
    const uint32_t tot = totals[t];  // kBlock == kRadix
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) woff += (w < wave) ? wave_tot[w] : 0u;
    digit_base[t] = woff + inc - tot + offsets[(uint64_t)t * groups + blockIdx.x];
    __syncthreads();
  }

  const uint64_t begin = (uint64_t)blockIdx.x * chunk;
  uint64_t end = begin + chunk;
  if (end > n) end = n;

  for (uint64_t base = begin; base < end; base += kTile) {
    const uint32_t tile_n = (uint32_t)((end - base) < (uint64_t)kTile ? (end - base) : (uint64_t)kTile);
#pragma unroll
    for (int w = 0; w < kWaves; ++w) whist[w][t] = 0;
    __syncthreads();

    KeyT key[kKpt];
    uint16_t lpos[kKpt];
    const uint32_t wbase = wave * 64 * kKpt + lane;
#pragma unroll
    for (int i = 0; i < kKpt; ++i) {
      const uint32_t li = wbase + i * 64;
      const bool valid = li < tile_n;
      key[i] = valid ? keys_in[base + li] : (KeyT)0;
      const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
      uint64_t peers = __ballot(valid);
      for (int b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
      }
      const uint32_t rank_in = __popcll(peers & lane_lt);
      const uint32_t cnt = __popcll(peers);
      uint32_t pre = 0;
      if (valid) pre = whist[wave][d];
      __builtin_amdgcn_wave_barrier();
      if (valid && rank_in == 0) whist[wave][d] = pre + cnt;
      __builtin_amdgcn_wave_barrier();
      lpos[i] = (uint16_t)(pre + rank_in);
    }
    __syncthreads();
    // digit t: exclusive prefix over the waves, tile count, then exclusive scan over the digits
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      uint32_t v = whist[w][t];
      whist[w][t] = acc;
      acc += v;
    }
    tile_count[t] = acc;
    uint32_t inc = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) woff += (w < wave) ? wave_tot[w] : 0u;
    tile_start[t] = woff + inc - acc;
    __syncthreads();
    // local sorted position of every key; stage keys in LDS
#pragma unroll
    for (int i = 0; i < kKpt; ++i) {
      const uint32_t li = wbase + i * 64;
      if (li < tile_n) {
        const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
        const uint32_t p = tile_start[d] + whist[wave][d] + lpos[i];
        lpos[i] = (uint16_t)p;
        skeys[p] = key[i];
      }
    }
    __syncthreads();
    uint32_t gidx[kKpt];
#pragma unroll
    for (int j = 0; j < kKpt; ++j) {
      const uint32_t p = j * kBlock + t;
      if (p < tile_n) {
        const KeyT k = skeys[p];
        const uint32_t d = (uint32_t)(k >> shift) & mask;
        const uint32_t g = digit_base[d] + (p - tile_start[d]);
        gidx[j] = g;
        keys_out[g] = k;
      }
    }
    for (int w = 0; w < pl.nwords; ++w) {
      const uint32_t* __restrict__ src = pl.in[w];
      uint32_t* __restrict__ dst = pl.out[w];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kKpt; ++i) {
        const uint32_t li = wbase + i * 64;
        if (li < tile_n) stage32[lpos[i]] = src[base + li];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kKpt; ++j) {
        const uint32_t p = j * kBlock + t;
        if (p < tile_n) dst[gidx[j]] = stage32[p];
      }
    }
    __syncthreads();
    digit_base[t] += tile_count[t];
    // next iteration's first barrier orders this update before any use
  }
}

template <typename KeyT>
int radix_sort(pcv_ctx* ctx, KeyT* a, KeyT* b, uint64_t n, int begin_bit, int end_bit, PcvSortPayload* payload,
               void* scratch, bool* result_in_a) {
  *result_in_a = true;
  if (n == 0 || end_bit <= begin_bit) return PCV_OK;
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "radix sort: n must be < 2^32 - 1");
  SortGeom g = make_geom(n);
  uint32_t* hist = (uint32_t*)scratch;
  uint32_t* totals = hist + (size_t)kRadix * kMaxGroups;
  bool in_a = true;
  for (int shift = begin_bit; shift < end_bit; shift += 8) {
    int nbits = end_bit - shift < 8 ? end_bit - shift : 8;
    uint32_t mask = (1u << nbits) - 1u;
    KeyT* src = in_a ? a : b;
    KeyT* dst = in_a ? b : a;
    PayloadPtrs pl{};
    if (payload) {
      pl.nwords = payload->nwords;
      for (int w = 0; w < payload->nwords; ++w) {
        pl.in[w] = in_a ? payload->in[w] : payload->out[w];
        pl.out[w] = in_a ? payload->out[w] : payload->in[w];
      }
    }
    {
      PcvProf prof(ctx, sizeof(KeyT) == 8 ? PCV_K_SORT_UPSWEEP64 : PCV_K_SORT_UPSWEEP32);
      hipLaunchKernelGGL(upsweep_kernel<KeyT>, dim3(g.groups), dim3(kBlock), 0, ctx->stream, src, n, g.chunk,
                         g.groups, shift, mask, hist);
    }
    {
      PcvProf prof(ctx, PCV_K_SORT_SCAN);
      hipLaunchKernelGGL(scan_kernel, dim3(kRadix), dim3(256), 0, ctx->stream, hist, g.groups, totals);
    }
    {
      PcvProf prof(ctx, sizeof(KeyT) == 8 ? PCV_K_SORT_DOWNSWEEP64 : PCV_K_SORT_DOWNSWEEP32);
      hipLaunchKernelGGL(downsweep_kernel<KeyT>, dim3(g.groups), dim3(kBlock), 0, ctx->stream, src, dst, n, g.chunk,
                         g.groups, shift, nbits, hist, totals, pl);
    }
    in_a = !in_a;
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  *result_in_a = in_a;
  return PCV_OK;
}

}  // namespace

size_t pcv_sort_scratch_bytes(uint64_t n) { return ((size_t)kRadix * kMaxGroups + kRadix) * sizeof(uint32_t); }

int pcv_radix_sort_u64(pcv_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, int begin_bit, int end_bit,
                       PcvSortPayload* payload, void* scratch, bool* result_in_a) {
  return radix_sort<uint64_t>(ctx, keys_a, keys_b, n, begin_bit, end_bit, payload, scratch, result_in_a);
}
int pcv_radix_sort_u32(pcv_ctx* ctx, uint32_t* keys_a, uint32_t* keys_b, uint64_t n, int begin_bit, int end_bit,
                       PcvSortPayload* payload, void* scratch, bool* result_in_a) {
  return radix_sort<uint32_t>(ctx, keys_a, keys_b, n, begin_bit, end_bit, payload, scratch, result_in_a);
}
