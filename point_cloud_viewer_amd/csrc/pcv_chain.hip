// pcv_chain.hip — K1 aabb_reduce and K2 chain_keys for gfx950.
//
// K1 replaces find_bounding_box (reference src/octree/generation.rs:256-270, Aabb::grow aabb.rs:41-44).
// K2 replaces the per-point work of split()/split_node() (generation.rs:58-193): the octant digit of every
// level along the point's quantise->decode chain, which does not depend on the tree topology.
//
// Bounds: K1 is a pure HBM stream (24 B/point). K2 is f64-VALU bound by construction (two correctly rounded
// f64 divisions per coordinate per level; parity forbids reciprocals) — it reads 24 B and writes 8 B/point.
#include <cstring>

#include "pcv_chain_dev.h"

namespace {

constexpr int kAabbBlock = 256;

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// Each block reduces a grid-strided slice of the three coordinate streams; one 6-double partial per block.
__global__ __launch_bounds__(kAabbBlock) void aabb_partial_kernel(uint64_t n, const double* __restrict__ x,
                                                                   const double* __restrict__ y,
                                                                   const double* __restrict__ z,
                                                                   double* __restrict__ partial) {
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  const uint64_t stride = (uint64_t)gridDim.x * kAabbBlock * 2;
  // two points per lane per iteration: 16-byte loads, fully coalesced
  for (uint64_t i = ((uint64_t)blockIdx.x * kAabbBlock + threadIdx.x) * 2; i < n; i += stride) {
    if (i + 1 < n) {
      // streaming (nontemporal) loads: every byte is read once — 0.385 -> 0.335-0.36 ms for 100 M points (6.7-7.2 TB/s),
      // A B A B in one call (the same loads in the chain pass, the count and the record sort gained nothing)
      typedef double d2v __attribute__((ext_vector_type(2)));
      const d2v ax = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(x + i));
      const d2v ay = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(y + i));
      const d2v az = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(z + i));
      const double2 vx = make_double2(ax.x, ax.y), vy = make_double2(ay.x, ay.y), vz = make_double2(az.x, az.y);
      lo[0] = fmin(lo[0], fmin(vx.x, vx.y));
      hi[0] = fmax(hi[0], fmax(vx.x, vx.y));
      lo[1] = fmin(lo[1], fmin(vy.x, vy.y));
      hi[1] = fmax(hi[1], fmax(vy.x, vy.y));
      lo[2] = fmin(lo[2], fmin(vz.x, vz.y));
      hi[2] = fmax(hi[2], fmax(vz.x, vz.y));
    } else {
      lo[0] = fmin(lo[0], x[i]);
      hi[0] = fmax(hi[0], x[i]);
      lo[1] = fmin(lo[1], y[i]);
      hi[1] = fmax(hi[1], y[i]);
      lo[2] = fmin(lo[2], z[i]);
      hi[2] = fmax(hi[2], z[i]);
    }
  }
  __shared__ double red[kAabbBlock / 64][6];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double l = wave_min(lo[a]), h = wave_max(hi[a]);
    if (lane == 0) {
      red[wave][a] = l;
      red[wave][3 + a] = h;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = red[0][threadIdx.x];
    for (int w = 1; w < kAabbBlock / 64; ++w)
      v = threadIdx.x < 3 ? fmin(v, red[w][threadIdx.x]) : fmax(v, red[w][threadIdx.x]);
    partial[(uint64_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(256) void aabb_final_kernel(int nblocks, const double* __restrict__ partial,
                                                          double* __restrict__ out6) {
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  __shared__ double red[4][6];
  double v[6] = {inf, inf, inf, -inf, -inf, -inf};
  for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double q = partial[(uint64_t)b * 6 + a];
      v[a] = a < 3 ? fmin(v[a], q) : fmax(v[a], q);
    }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double r = a < 3 ? wave_min(v[a]) : wave_max(v[a]);
    if (lane == 0) red[wave][a] = r;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double r = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) r = threadIdx.x < 3 ? fmin(r, red[w][threadIdx.x]) : fmax(r, red[w][threadIdx.x]);
    out6[threadIdx.x] = r;
  }
}

// K2: one point per lane; the level loop is wave-uniform (levels, edges and encodings are kernel
// arguments in SGPRs), so the encoding switch is a scalar branch. `stride` > 1 evaluates a strided sample of the
// input (depth probe); KeyT = u32 stores the top 10 levels only (key >> 33).
template <typename KeyT>
__global__ __launch_bounds__(256) void chain_keys_kernel(PcvLevels lv, uint64_t n, uint64_t stride, uint32_t clump_shift,
                                                          const double* __restrict__ x, const double* __restrict__ y,
                                                          const double* __restrict__ z, PcvRouted routed,
                                                          KeyT* __restrict__ keys, uint4* __restrict__ zero, uint32_t zero_vecs) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  // the counters of the sort that takes these keys next (pcv_sort_keys_onesweep) are cleared on the way: no launch of their own
  for (uint64_t j = i; j < zero_vecs; j += (uint64_t)gridDim.x * 256) zero[j] = make_uint4(0u, 0u, 0u, 0u);
  if (i >= n) return;
  // sample i of a strided sample taken in clumps of 2^clump_shift consecutive points (same density: one clump every
  // stride x clump points): a lone 8-byte coordinate costs a whole cache line, a clump uses the line it fetches
  const uint64_t src = (((i >> clump_shift) * stride) << clump_shift) + (i & ((1ull << clump_shift) - 1ull));
  double px, py, pz, mx, my, mz;
  double cx = 0, cy = 0, cz = 0;
  uint32_t d1;
  const int k0 = pcv_chain_start(lv, routed, x, y, z, src, px, py, pz, mx, my, mz, cx, cy, cz, d1);
  uint64_t key = (uint64_t)d1 << (3 * (PCV_MAX_KEY_LEVELS - 1));
  if (lv.fast_ok && pcv_point_is_tame(px, py, pz)) {
    for (int k = k0; k <= lv.nlevels; ++k) {
      const uint32_t d = pcv_chain_level<false>(lv.enc[k], lv.edge[k - 1], lv.edge[k], PcvRecip{lv.inv_edge[k], lv.inv_edge_lo[k]}, px, py, pz, mx, my, mz, cx, cy, cz);
      key |= (uint64_t)d << (3 * (PCV_MAX_KEY_LEVELS - k));
    }
  } else {
    for (int k = k0; k <= lv.nlevels; ++k) {
      const uint32_t d = pcv_chain_level<true>(lv.enc[k], lv.edge[k - 1], lv.edge[k], PcvRecip{lv.inv_edge[k], lv.inv_edge_lo[k]}, px, py, pz, mx, my, mz, cx, cy, cz);
      key |= (uint64_t)d << (3 * (PCV_MAX_KEY_LEVELS - k));
    }
  }
  keys[i] = sizeof(KeyT) == 8 ? (KeyT)key : (KeyT)(key >> 33);
}

// Deep trees (more than PCV_MAX_KEY_LEVELS levels): the same chain, digits in two key words.
struct DeepWords {
  uint32_t* w[4];
};
__global__ __launch_bounds__(256) void chain_keys_deep_kernel(PcvLevels lv, uint64_t n, const double* __restrict__ x,
                                                               const double* __restrict__ y, const double* __restrict__ z,
                                                               PcvRouted routed, DeepWords out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double px, py, pz, mx, my, mz;
  double cx = 0, cy = 0, cz = 0;
  uint32_t d1;
  const int k0 = pcv_chain_start(lv, routed, x, y, z, i, px, py, pz, mx, my, mz, cx, cy, cz, d1);
  uint64_t hi = (uint64_t)d1 << (3 * (PCV_MAX_KEY_LEVELS - 1)), lo = 0;
  for (int k = k0; k <= lv.nlevels; ++k) {  // rare path: always the guarded variant
    const uint64_t d = pcv_chain_level<true>(lv.enc[k], lv.edge[k - 1], lv.edge[k], PcvRecip{lv.inv_edge[k], lv.inv_edge_lo[k]}, px, py, pz, mx, my, mz, cx, cy, cz);
    if (k <= PCV_MAX_KEY_LEVELS)
      hi |= d << (3 * (PCV_MAX_KEY_LEVELS - k));
    else
      lo |= d << (3 * (2 * PCV_MAX_KEY_LEVELS - k));
  }
  out.w[0][i] = (uint32_t)(hi >> 32);
  out.w[1][i] = (uint32_t)hi;
  out.w[2][i] = (uint32_t)(lo >> 32);
  out.w[3][i] = (uint32_t)lo;
}
__global__ __launch_bounds__(256) void combine_words_kernel(uint64_t n, DeepWords in, uint64_t* __restrict__ hi,
                                                             uint64_t* __restrict__ lo) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  hi[i] = ((uint64_t)in.w[0][i] << 32) | in.w[1][i];
  lo[i] = ((uint64_t)in.w[2][i] << 32) | in.w[3][i];
}

// Depth probe on a sorted sample: if two keys `gap` positions apart share their first l digits, the level-l node
// holding them has more than `gap` sample points. The maximum such l over the sample bounds the deepest node that
// the full input will have to split.
__global__ __launch_bounds__(256) void depth_probe_kernel(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t gap,
                                                           uint32_t* __restrict__ max_shared_levels) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t l = 0;
  if (i + gap < n) {
    const uint64_t diff = sorted[i] ^ sorted[i + gap];
    // keys use bits 62..0; clz counts from bit 63
    const uint32_t lead = diff ? (uint32_t)__clzll((long long)diff) : 64u;
    l = lead >= 1 ? (lead - 1) / 3 : 0;
    if (l > PCV_MAX_KEY_LEVELS) l = PCV_MAX_KEY_LEVELS;
  }
  for (int o = 32; o > 0; o >>= 1) l = max(l, (uint32_t)__shfl_xor((int)l, o, 64));
  // one atomic per workgroup, and only when it would raise the maximum (a plain read first keeps the waves off one address)
  __shared__ uint32_t wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (m > *(volatile uint32_t*)max_shared_levels) atomicMax(max_shared_levels, m);
  }
}

// Multi-GPU routing: the bucket of a point is its level-1 and level-2 octant digit (the first two steps of
// the K2 chain, so bit-identical to what the owner's build computes again), bucket = 8 * d1 + d2. The 64 counts
// are what the ranks all-reduce to decide the global top of the tree and to bin-pack buckets onto ranks.
__global__ __launch_bounds__(256) void route_bucket_kernel(PcvLevels lv, uint64_t n, const double* __restrict__ x,
                                                            const double* __restrict__ y, const double* __restrict__ z,
                                                            uint32_t* __restrict__ bucket,
                                                            unsigned long long* __restrict__ counts /* [64] */,
                                                            uint32_t* __restrict__ st_orgb, uint32_t* __restrict__ st_cx,
                                                            uint32_t* __restrict__ st_cy, uint32_t* __restrict__ st_cz,
                                                            const uint8_t* __restrict__ color, uint32_t color_stride) {
  __shared__ uint32_t hist[64];
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256; i0 < n; i0 += stride) {
    const uint64_t i = i0 + threadIdx.x;
    const bool in = i < n;
    uint32_t b = 0;
    if (in) {
      double px = x[i], py = y[i], pz = z[i];
      double mx = lv.root_min[0], my = lv.root_min[1], mz = lv.root_min[2];
      double cx, cy, cz;
      for (int k = 1; k <= lv.nlevels; ++k) {  // nlevels <= 2 here; always the guarded (exact for any input) variant
        b = (b << 3) | pcv_chain_level<true>(lv.enc[k], lv.edge[k - 1], lv.edge[k], PcvRecip{lv.inv_edge[k], lv.inv_edge_lo[k]}, px, py, pz, mx, my, mz, cx, cy, cz);
        if (k == 1 && st_orgb) {  // the level-1 state that crosses the exchange instead of the raw coordinates
          const uint8_t* c = color + i * color_stride;
          st_orgb[i] = b | ((uint32_t)c[0] << 8) | ((uint32_t)c[1] << 16) | ((uint32_t)c[2] << 24);
          st_cx[i] = __float_as_uint((float)cx);  // value domain -> Float32 bit pattern (exact: cx is a float value)
          st_cy[i] = __float_as_uint((float)cy);
          st_cz[i] = __float_as_uint((float)cz);
        }
      }
      if (lv.nlevels < 2) b <<= 3;
      bucket[i] = b;
    }
    // wave-aggregated histogram: lanes with the same bucket elect one leader (match-any by 6 ballots)
    uint64_t peers = __ballot(in);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
      const uint64_t m = __ballot((b >> bit) & 1u);
      peers &= ((b >> bit) & 1u) ? m : ~m;
    }
    if (in && (peers & ((1ull << lane) - 1ull)) == 0) atomicAdd(&hist[b], (uint32_t)__popcll(peers));
  }
  __syncthreads();
  if (threadIdx.x < 64 && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// Stable partition of the point planes by owner (<= 8 destinations), writing every owner's rows to its own
// destination pointers (send buffers of the other ranks, the receive buffer for the own rank): count per tile,
// scan per owner, scatter. Input order is preserved inside every destination.
constexpr int kPartTile = 4096;  // 256 lanes x 16 rows, wave-striped like the radix sort

__global__ __launch_bounds__(256) void partition_count_kernel(uint64_t n, const uint32_t* __restrict__ owner, uint32_t world,
                                                               uint32_t ntiles, uint32_t* __restrict__ tile_counts /* [world][ntiles] */,
                                                               const uint8_t* __restrict__ remap /* [64] bucket -> rank, or null */) {
  __shared__ uint32_t cnt[8];
  __shared__ uint8_t rank_of[64];
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
  if (threadIdx.x < 64) rank_of[threadIdx.x] = remap ? remap[threadIdx.x] : (uint8_t)threadIdx.x;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kPartTile;
  uint32_t local[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) {
    const uint64_t idx = base + (uint64_t)i * 256 + threadIdx.x;
    if (idx < n) {
      const uint32_t o = rank_of[owner[idx] & 63u];
#pragma unroll
      for (int k = 0; k < 8; ++k) local[k] += (o == (uint32_t)k) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t v = local[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cnt[k], v);
  }
  __syncthreads();
  if (threadIdx.x < world) tile_counts[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

// one workgroup per owner: exclusive scan of its ntiles counters in place
__global__ __launch_bounds__(1024) void partition_scan_kernel(uint32_t* __restrict__ tile_counts, uint32_t ntiles) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t running;
  uint32_t* row = tile_counts + (uint64_t)blockIdx.x * ntiles;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t base = 0; base < ntiles; base += 1024) {
    const uint32_t idx = base + threadIdx.x;
    const uint32_t v = idx < ntiles ? row[idx] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      woff += (w < wave) ? wave_tot[w] : 0u;
      tot += wave_tot[w];
    }
    if (idx < ntiles) row[idx] = running + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) running += tot;
    __syncthreads();
  }
}

constexpr int kMaxPlanes = 8;
struct PartPlanes {
  const uint8_t* src[kMaxPlanes];
  uint32_t elem[kMaxPlanes];  // bytes per row: 1, 2, 3, 4, 8 or 16
  uint8_t* dst[8][kMaxPlanes];
  int nplanes;
};

__device__ __forceinline__ void copy_row(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, uint32_t elem) {
  switch (elem) {  // wave-uniform
    case 8: *reinterpret_cast<uint64_t*>(d) = *reinterpret_cast<const uint64_t*>(s); break;
    case 4: *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(s); break;
    case 2: *reinterpret_cast<uint16_t*>(d) = *reinterpret_cast<const uint16_t*>(s); break;
    case 16: *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s); break;
    default:
      for (uint32_t b = 0; b < elem; ++b) d[b] = s[b];
  }
}

__global__ __launch_bounds__(256) void partition_scatter_kernel(uint64_t n, const uint32_t* __restrict__ owner, uint32_t world,
                                                                 uint32_t ntiles, const uint32_t* __restrict__ tile_base,
                                                                 PartPlanes pl, const uint8_t* __restrict__ remap) {
  __shared__ uint32_t wave_cnt[4][8];
  __shared__ uint8_t* sdst[8][kMaxPlanes];  // per-lane owner indexes the pointer table: LDS lookup, not a private copy
  __shared__ uint8_t rank_of[64];
  if (threadIdx.x < 8 * kMaxPlanes) sdst[threadIdx.x / kMaxPlanes][threadIdx.x % kMaxPlanes] = pl.dst[threadIdx.x / kMaxPlanes][threadIdx.x % kMaxPlanes];
  if (threadIdx.x < 64) rank_of[threadIdx.x] = remap ? remap[threadIdx.x] : (uint8_t)threadIdx.x;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  const uint64_t base = (uint64_t)blockIdx.x * kPartTile + (uint64_t)wave * 1024 + lane;
  uint32_t own[16];
  uint32_t wcount[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint64_t idx = base + (uint64_t)i * 64;
    own[i] = idx < n ? (uint32_t)rank_of[owner[idx] & 63u] : 0xffu;
    for (uint32_t k = 0; k < world; ++k) wcount[k] += (uint32_t)__popcll(__ballot(own[i] == k));
  }
  if (lane == 0)
    for (uint32_t k = 0; k < world; ++k) wave_cnt[wave][k] = wcount[k];
  __syncthreads();
  uint32_t run[8];  // next free row of each owner for this wave
  for (uint32_t k = 0; k < world; ++k) {
    uint32_t r = tile_base[(uint64_t)k * ntiles + blockIdx.x];
    for (int w = 0; w < wave; ++w) r += wave_cnt[w][k];
    run[k] = r;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint64_t idx = base + (uint64_t)i * 64;
    const uint32_t o = own[i];
    uint32_t pos = 0;
    for (uint32_t k = 0; k < world; ++k) {
      const uint64_t m = __ballot(o == k);
      if (o == k) pos = run[k] + (uint32_t)__popcll(m & lane_lt);
      run[k] += (uint32_t)__popcll(m);
    }
    if (idx < n) {
      for (int p = 0; p < pl.nplanes; ++p) {
        const uint32_t e = pl.elem[p];
        copy_row(sdst[o][p] + (uint64_t)pos * e, pl.src[p] + idx * e, e);
      }
    }
  }
}

// ---- routing in two passes (round 4) ---------------------------------------------------------------------------------------
// route_bucket + partition_count + partition_scatter above move 47 + 4 + 36 bytes per point: the first writes the 16-byte
// level-1 state of every point next to its bucket, the last reads it back to scatter it. Here the state never exists in
// input order:
//   pass 1 (route_plan):    coordinates -> bucket BYTE per point + the bucket histogram of every tile of 4 096 points + the 64
//                           counts the ranks all-gather (24 B read + 1 B written per point);
//   pass 2 (route_scatter): the plan (bucket -> owner) is known; owner counts per tile = sums of the tile's histogram, scanned;
//                           then coordinates + colour + bucket byte -> the level-1 state, computed in registers (the octant
//                           digit is the bucket's upper three bits, so the step is one `min += bit * edge` and one Float32
//                           encode per coordinate — the operations pass 1 ran, on the same operands) and stored straight at
//                           the point's stable place in its owner's send buffer (28 B read + 16 B written per point).
// 69 instead of 87 bytes per point, no partition_count pass.
constexpr int kRouteTile = kPartTile;  // 4 096 points: 256 lanes x 16 rows, wave-striped

__global__ __launch_bounds__(256) void route_plan_kernel(PcvLevels lv, uint64_t n, const double* __restrict__ x,
                                                          const double* __restrict__ y, const double* __restrict__ z,
                                                          uint8_t* __restrict__ bucket, uint16_t* __restrict__ tile_hist /* [tiles][64] */,
                                                          unsigned long long* __restrict__ counts /* [64] */,
                                                          bool octants_only /* PCV_ROUTE_OCTANTS_ONLY: bucket = d1 << 3 */) {
  __shared__ uint32_t hist[64];
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * kRouteTile + threadIdx.x;
  // round 5: the usual case — a tame table whose level 1 is Float32-coded with the digit of level 2 readable off the level-1
  // codes (PcvLevels::digit_mode, the rule the chain pass follows) — takes ONE unguarded level step and three compares per
  // point instead of two guarded steps: the second digit is v > 1/2 per coordinate; a code of exactly 1/2 (a tie of the
  // exact values) is decided by the comparison against the centre, as in the chain pass
  const bool fast = lv.fast_ok && lv.nlevels >= 2 && lv.enc[1] == PCV_ENC_FLOAT32 && lv.digit_mode[1] == 2u;
#pragma unroll 4
  for (int k = 0; k < kRouteTile / 256; ++k) {
    const uint64_t i = base + (uint64_t)k * 256;
    if (i < n) {
      double px = x[i], py = y[i], pz = z[i];
      double mx = lv.root_min[0], my = lv.root_min[1], mz = lv.root_min[2];
      double cx, cy, cz;
      uint32_t b = 0;
      if (octants_only) {
        // round 6: ownership by root octant (BASELINE north_star: the top-3-bit prefix) needs the level-1 digit alone — three
        // comparisons against the root cube's centre (node.rs:34-42), for any input: the pass is a plain stream of the coordinates
        b = pcv_chain_bits(lv.edge[0], px, py, pz, mx, my, mz).digit() << 3;
      } else if (fast && pcv_point_is_tame(px, py, pz)) {
        const PcvOctBits b1 = pcv_chain_bits(lv.edge[0], px, py, pz, mx, my, mz);
        pcv_chain_apply_bits_t<PCV_ENC_FLOAT32, false>(b1, lv.edge[1], PcvRecip{lv.inv_edge[1], lv.inv_edge_lo[1]}, px, py, pz, mx, my, mz, cx, cy, cz);
        PcvOctBits b2 = pcv_bits_from_codes(0.5, cx, cy, cz);
        if (__builtin_expect(pcv_f32_code_tie(cx, cy, cz), 0)) b2 = pcv_chain_bits(lv.edge[1], px, py, pz, mx, my, mz);
        b = (b1.digit() << 3) | b2.digit();
      } else {
      for (int l = 1; l <= lv.nlevels; ++l)  // nlevels <= 2 here; the guarded (exact for any input) variant
        b = (b << 3) | pcv_chain_level<true>(lv.enc[l], lv.edge[l - 1], lv.edge[l], PcvRecip{lv.inv_edge[l], lv.inv_edge_lo[l]}, px, py, pz, mx, my, mz, cx, cy, cz);
      }
      if (lv.nlevels < 2 && !octants_only) b <<= 3;
      bucket[i] = (uint8_t)b;
      atomicAdd(&hist[b], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t v = hist[threadIdx.x];
    tile_hist[(uint64_t)blockIdx.x * 64 + threadIdx.x] = (uint16_t)v;  // <= 4 096
    if (v) atomicAdd(&counts[threadIdx.x], (unsigned long long)v);
  }
}

// pass 1 when ownership goes by root octant (PCV_ROUTE_OCTANTS_ONLY, round 6): the level-1 digit alone — three comparisons against
// the root cube's centre (node.rs:34-42), exact for any input — so the pass is a plain stream: every lane takes 4 x 4 consecutive
// points (16-byte loads, one dword of bucket bytes per store), the tile's 8 counts are wave ballots (the shared histogram's
// LDS atomics would all land on 8 addresses: 0.62 ms measured for the pass that way, the cost of the full level step).
__global__ __launch_bounds__(256) void route_octants_kernel(PcvLevels lv, uint64_t n, const double* __restrict__ x,
                                                             const double* __restrict__ y, const double* __restrict__ z,
                                                             uint8_t* __restrict__ bucket, uint16_t* __restrict__ tile_hist /* [tiles][64] */,
                                                             unsigned long long* __restrict__ counts /* [64] */) {
  __shared__ uint32_t wave_cnt[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double e0 = lv.edge[0];
  const double mx = lv.root_min[0], my = lv.root_min[1], mz = lv.root_min[2];
  const double cx = (mx + (mx + e0)) / 2.0, cy = (my + (my + e0)) / 2.0, cz = (mz + (mz + e0)) / 2.0;  // pcv_chain_bits' centre
  uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform
  const uint64_t tile0 = (uint64_t)blockIdx.x * kRouteTile;
#pragma unroll
  for (int k = 0; k < kRouteTile / 1024; ++k) {
    const uint64_t i = tile0 + (uint64_t)k * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t d[4] = {0xffu, 0xffu, 0xffu, 0xffu};
    if (i + 4 <= n) {  // (the arrays come from the pool or are checked for 16-byte alignment by the caller)
      const double2 xa = *reinterpret_cast<const double2*>(x + i), xb = *reinterpret_cast<const double2*>(x + i + 2);
      const double2 ya = *reinterpret_cast<const double2*>(y + i), yb = *reinterpret_cast<const double2*>(y + i + 2);
      const double2 za = *reinterpret_cast<const double2*>(z + i), zb = *reinterpret_cast<const double2*>(z + i + 2);
      d[0] = (xa.x > cx ? 4u : 0u) | (ya.x > cy ? 2u : 0u) | (za.x > cz ? 1u : 0u);
      d[1] = (xa.y > cx ? 4u : 0u) | (ya.y > cy ? 2u : 0u) | (za.y > cz ? 1u : 0u);
      d[2] = (xb.x > cx ? 4u : 0u) | (yb.x > cy ? 2u : 0u) | (zb.x > cz ? 1u : 0u);
      d[3] = (xb.y > cx ? 4u : 0u) | (yb.y > cy ? 2u : 0u) | (zb.y > cz ? 1u : 0u);
      *reinterpret_cast<uint32_t*>(bucket + i) = (d[0] << 3) | (d[1] << 11) | (d[2] << 19) | (d[3] << 27);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < n) {
          d[j] = (x[i + j] > cx ? 4u : 0u) | (y[i + j] > cy ? 2u : 0u) | (z[i + j] > cz ? 1u : 0u);
          bucket[i + j] = (uint8_t)(d[j] << 3);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (uint32_t o = 0; o < 8; ++o) cnt[o] += (uint32_t)__popcll(__ballot(d[j] == o));
  }
  if (lane == 0)
#pragma unroll
    for (int o = 0; o < 8; ++o) wave_cnt[wave][o] = cnt[o];
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t o = threadIdx.x >> 3;
    const uint32_t v = (threadIdx.x & 7u) == 0u ? wave_cnt[0][o] + wave_cnt[1][o] + wave_cnt[2][o] + wave_cnt[3][o] : 0u;
    tile_hist[(uint64_t)blockIdx.x * 64 + threadIdx.x] = (uint16_t)v;  // <= 4 096
  }
  // (no global counters here: 24 414 workgroups adding to the same 8 addresses is what held the first form of this pass at
  // 0.58 ms; the 64 counts are the column sums of the tile histograms, route_hist_sum_kernel)
}
// counts[b] += sum over this workgroup's tiles of tile_hist[tile][b]: 128 workgroups x 16 rows of tiles per step, then 64 atomics
// per workgroup (counts is zeroed by the caller)
constexpr int kHistSumBlocks = 128;
__global__ __launch_bounds__(1024) void route_hist_sum_kernel(const uint16_t* __restrict__ tile_hist, uint32_t ntiles,
                                                               unsigned long long* __restrict__ counts) {
  __shared__ unsigned long long part[16][64];
  const uint32_t col = threadIdx.x & 63u, r0 = threadIdx.x >> 6;
  const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const uint32_t t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  unsigned long long acc = 0;
  for (uint32_t t = t0 + r0; t < t1; t += 16) acc += tile_hist[(uint64_t)t * 64 + col];
  part[r0][col] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    unsigned long long v = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += part[r][threadIdx.x];
    if (v) atomicAdd(&counts[threadIdx.x], v);
  }
}

// owner counts per tile from the tiles' bucket histograms: tile_counts[owner][tile]
__global__ __launch_bounds__(256) void route_owner_counts_kernel(const uint16_t* __restrict__ tile_hist, uint32_t ntiles,
                                                                  const uint8_t* __restrict__ remap, uint32_t world,
                                                                  uint32_t* __restrict__ tile_counts) {
  __shared__ uint8_t rank_of[64];
  if (threadIdx.x < 64) rank_of[threadIdx.x] = remap[threadIdx.x];
  __syncthreads();
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint4* h4 = reinterpret_cast<const uint4*>(tile_hist + (uint64_t)t * 64);  // 128 bytes per tile
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint4 v = h4[q];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t b0 = (uint32_t)(q * 8 + e * 2);
      const uint32_t r0 = rank_of[b0], r1 = rank_of[b0 + 1];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] += (r0 == (uint32_t)k ? (w[e] & 0xffffu) : 0u) + (r1 == (uint32_t)k ? (w[e] >> 16) : 0u);
    }
  }
  for (uint32_t k = 0; k < world; ++k) tile_counts[(uint64_t)k * ntiles + t] = c[k];
}

struct RouteDst {
  uint32_t* orgb[8];
  uint32_t* cx[8];
  uint32_t* cy[8];
  uint32_t* cz[8];
  float* inten[8];
};

__global__ __launch_bounds__(256) void route_scatter_kernel(PcvLevels lv, uint64_t n, const double* __restrict__ x,
                                                             const double* __restrict__ y, const double* __restrict__ z,
                                                             const uint8_t* __restrict__ color, uint32_t color_stride,
                                                             const float* __restrict__ intensity, const uint8_t* __restrict__ bucket,
                                                             const uint8_t* __restrict__ remap, uint32_t world, uint32_t ntiles,
                                                             const uint32_t* __restrict__ tile_base, RouteDst dst) {
  __shared__ uint32_t wave_cnt[4][8];
  __shared__ uint8_t rank_of[64];
  __shared__ uint32_t* sp[5][8];  // per-lane owner indexes the pointer tables: LDS lookup, not a private copy
  if (threadIdx.x < 64) rank_of[threadIdx.x] = remap[threadIdx.x];
  if (threadIdx.x < 8) {
    sp[0][threadIdx.x] = dst.orgb[threadIdx.x];
    sp[1][threadIdx.x] = dst.cx[threadIdx.x];
    sp[2][threadIdx.x] = dst.cy[threadIdx.x];
    sp[3][threadIdx.x] = dst.cz[threadIdx.x];
    sp[4][threadIdx.x] = reinterpret_cast<uint32_t*>(dst.inten[threadIdx.x]);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  const uint64_t base = (uint64_t)blockIdx.x * kRouteTile + (uint64_t)wave * 1024 + lane;
  uint32_t own[16], d1[16];
  uint32_t wcount[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint64_t idx = base + (uint64_t)i * 64;
    const uint32_t b = idx < n ? (uint32_t)bucket[idx] : 0xffu;
    own[i] = b < 64u ? (uint32_t)rank_of[b] : 0xffu;
    d1[i] = (b >> 3) & 7u;
    for (uint32_t k = 0; k < world; ++k) wcount[k] += (uint32_t)__popcll(__ballot(own[i] == k));
  }
  if (lane == 0)
    for (uint32_t k = 0; k < world; ++k) wave_cnt[wave][k] = wcount[k];
  __syncthreads();
  uint32_t run[8];  // next free row of each owner for this wave
  for (uint32_t k = 0; k < world; ++k) {
    uint32_t r = tile_base[(uint64_t)k * ntiles + blockIdx.x];
    for (int w = 0; w < wave; ++w) r += wave_cnt[w][k];
    run[k] = r;
  }
  const double e1 = lv.edge[1];
  const PcvRecip r1{lv.inv_edge[1], lv.inv_edge_lo[1]};
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const uint64_t idx = base + (uint64_t)i * 64;
    const uint32_t o = own[i];
    uint32_t pos = 0;
    for (uint32_t k = 0; k < world; ++k) {
      const uint64_t m = __ballot(o == k);
      if (o == k) pos = run[k] + (uint32_t)__popcll(m & lane_lt);
      run[k] += (uint32_t)__popcll(m);
    }
    if (idx < n) {
      // the level-1 step of the chain for a known digit (pcv_chain_coord_t: min += bit * edge, Float32 encode, guarded variant)
      const uint32_t d = d1[i];
      const double mx = pcv_step_min(lv.root_min[0], (d & 4u) != 0u, e1), my = pcv_step_min(lv.root_min[1], (d & 2u) != 0u, e1),
                   mz = pcv_step_min(lv.root_min[2], (d & 1u) != 0u, e1);
      // (round 5: a tame point in a tame table takes the unguarded quotient — the same value, one range test instead of three)
      const double px = x[idx], py = y[idx], pz = z[idx];
      double cx, cy, cz;
      if (lv.fast_ok && pcv_point_is_tame(px, py, pz)) {
        cx = pcv_encode_val<PCV_ENC_FLOAT32, false>(px, mx, e1, r1), cy = pcv_encode_val<PCV_ENC_FLOAT32, false>(py, my, e1, r1),
        cz = pcv_encode_val<PCV_ENC_FLOAT32, false>(pz, mz, e1, r1);
      } else {
        cx = pcv_encode_val<PCV_ENC_FLOAT32, true>(px, mx, e1, r1), cy = pcv_encode_val<PCV_ENC_FLOAT32, true>(py, my, e1, r1),
        cz = pcv_encode_val<PCV_ENC_FLOAT32, true>(pz, mz, e1, r1);
      }
      const uint8_t* c = color + idx * color_stride;
      sp[0][o][pos] = d | ((uint32_t)c[0] << 8) | ((uint32_t)c[1] << 16) | ((uint32_t)c[2] << 24);
      sp[1][o][pos] = __float_as_uint((float)cx);  // value domain -> Float32 bit pattern (exact: cx is a float value)
      sp[2][o][pos] = __float_as_uint((float)cy);
      sp[3][o][pos] = __float_as_uint((float)cz);
      if (intensity) reinterpret_cast<float*>(sp[4][o])[pos] = intensity[idx];
    }
  }
}

// Division self-test: pcv_div_code against IEEE division for every code and both divisors (exhaustive), and
// pcv_div_const against IEEE division for pseudo-random numerators (full exponent/mantissa spread, plus values
// straddling rounding boundaries of the quotient) over a list of divisors.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void selftest_division_kernel(const double* __restrict__ divisors, int ndiv,
                                                                 uint64_t samples_per_divisor,
                                                                 unsigned long long* __restrict__ mismatches) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long bad = 0;
  if (gid < 65536) {
    const double v = (double)(uint32_t)gid;
    bad += __double_as_longlong(pcv_div_code(v, PCV_RECIP_65535)) != __double_as_longlong(v / 65535.0);
    if (gid < 256) bad += __double_as_longlong(pcv_div_code(v, PCV_RECIP_255)) != __double_as_longlong(v / 255.0);
  }
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (int d = 0; d < ndiv; ++d) {
    const double e = divisors[d], yh = (e >= 0x1p-100 && e <= 0x1p+100) ? 1.0 / e : 0.0;
    const PcvRecip y{yh, yh != 0.0 ? __fma_rn(-e, yh, 1.0) / e : 0.0};  // as pcv_make_levels builds it on the host
    for (uint64_t i = gid; i < samples_per_divisor; i += stride) {
      const uint64_t h = mix64(i * 0x100000001B3ull + (uint64_t)d);
      double x;
      switch (h & 3) {
        case 0:  // arbitrary bit pattern (any exponent, inf/nan/denormals included)
          x = __longlong_as_double((long long)mix64(h));
          break;
        case 1: {  // metres-scale magnitudes, random mantissa
          const int ex = (int)((h >> 8) % 80) - 40;
          x = ldexp(1.0 + (double)(mix64(h) >> 12) * 0x1p-52, ex) * ((h & 4) ? -1.0 : 1.0);
          break;
        }
        case 2: {  // numerators whose quotient sits next to a rounding boundary: x = RN(q * e) +- few ulp
          const double q = 1.0 + (double)(mix64(h) >> 12) * 0x1p-52;
          x = q * e;
          x = __longlong_as_double(__double_as_longlong(x) + (long long)((h >> 4) % 5) - 2);
          break;
        }
        default:  // small integers and their neighbours (codes, cell counts)
          x = (double)(int64_t)((h >> 3) % 200001) - 100000.0 + (double)((h >> 40) & 3) * 0.25;
      }
      const double a = pcv_div_const(x, e, y), b = x / e;
      const bool same = __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
      bad += same ? 0 : 1;
    }
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

}  // namespace

extern "C" int pcv_selftest_division(pcv_ctx* ctx, const double* divisors, int ndiv, uint64_t samples_per_divisor,
                                     uint64_t* mismatches) {
  if (!ctx) return PCV_E_INVALID;
  if (!mismatches || (ndiv && !divisors) || ndiv < 0) return ctx->fail(PCV_E_INVALID, "bad argument");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  double* dd;
  unsigned long long* dm;
  int rc;
  if ((rc = sc.get(&dd, (size_t)(ndiv ? ndiv : 1))) || (rc = sc.get(&dm, 1))) return rc;
  if (ndiv) PCV_HIP_CHECK(ctx, hipMemcpyAsync(dd, divisors, sizeof(double) * ndiv, hipMemcpyHostToDevice, ctx->stream));
  PCV_HIP_CHECK(ctx, hipMemsetAsync(dm, 0, 8, ctx->stream));
  hipLaunchKernelGGL(selftest_division_kernel, dim3(2048), dim3(256), 0, ctx->stream, dd, ndiv, samples_per_divisor, dm);
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, dm, 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  *mismatches = ctx->mailbox[0];
  return PCV_OK;
}

extern "C" int pcv_route_buckets(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, uint32_t* bucket,
                                 uint64_t counts[64], const pcv_route_state* state) {
  if (!ctx) return PCV_E_INVALID;
  if (!params || !points || !counts) return ctx->fail(PCV_E_INVALID, "null argument");
  for (int b = 0; b < 64; ++b) counts[b] = 0;
  if (points->n == 0) return PCV_OK;  // an empty input slice (a rank without points) is fine
  if (!bucket) return ctx->fail(PCV_E_INVALID, "bucket is null");
  if (points->mem != PCV_MEM_DEVICE) return ctx->fail(PCV_E_INVALID, "pcv_route_buckets works on device-resident points");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvLevels lv;
  int max_level;
  pcv_make_levels(params->bbox_min, params->bbox_max, params->resolution, 2, &lv, &max_level, nullptr, nullptr);
  if (state) {
    if (!state->oct_rgb || !state->cx || !state->cy || !state->cz) return ctx->fail(PCV_E_INVALID, "null plane in pcv_route_state");
    if (!points->color || (points->color_stride != 3 && points->color_stride != 4))
      return ctx->fail(PCV_E_INVALID, "the level-1 state packs the colour: points->color (stride 3 or 4) is required");
    if (lv.nlevels < 1 || lv.enc[1] != PCV_ENC_FLOAT32)
      return ctx->fail(PCV_E_INVALID, "the level-1 state is only defined for a Float32-encoded level 1: exchange raw coordinates");
  }
  PcvScratch sc(ctx);
  unsigned long long* d_counts;
  int rc;
  if ((rc = sc.get(&d_counts, 64))) return rc;
  PCV_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, 64 * 8, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_ROUTE_BUCKET);
    hipLaunchKernelGGL(route_bucket_kernel, dim3((unsigned)std::min<uint64_t>((points->n + 255) / 256, 8192)), dim3(256), 0,
                       ctx->stream, lv, points->n, points->x, points->y, points->z, bucket, d_counts,
                       state ? state->oct_rgb : nullptr, state ? state->cx : nullptr, state ? state->cy : nullptr,
                       state ? state->cz : nullptr, points->color, points->color_stride);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_counts, 64 * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < 64; ++b) counts[b] = ctx->mailbox[b];
  return PCV_OK;
}

extern "C" int pcv_partition_by_owner(pcv_ctx* ctx, uint64_t n, const uint32_t* owner, uint32_t world,
                                      const uint8_t* rank_of_bucket, uint32_t nplanes, const pcv_plane* planes,
                                      void* const* dst) {
  if (!ctx) return PCV_E_INVALID;
  if (!planes || !dst) return ctx->fail(PCV_E_INVALID, "null argument");
  if (world < 1 || world > 8) return ctx->fail(PCV_E_INVALID, "world must be 1..8");
  if (nplanes < 1 || nplanes > (uint32_t)kMaxPlanes) return ctx->fail(PCV_E_INVALID, "1..8 planes");
  if (n == 0) return PCV_OK;
  if (!owner) return ctx->fail(PCV_E_INVALID, "owner is null");
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 rows per call");
  if (rank_of_bucket)
    for (int b = 0; b < 64; ++b)
      if (rank_of_bucket[b] >= world) return ctx->fail(PCV_E_INVALID, "rank_of_bucket entry out of range");
  PartPlanes pl{};
  pl.nplanes = (int)nplanes;
  for (uint32_t p = 0; p < nplanes; ++p) {
    const uint32_t e = planes[p].elem_bytes;
    if (!planes[p].src || e == 0 || e > 16) return ctx->fail(PCV_E_INVALID, "plane: null source or row size outside 1..16 bytes");
    pl.src[p] = (const uint8_t*)planes[p].src;
    pl.elem[p] = e;
    for (uint32_t k = 0; k < world; ++k) pl.dst[k][p] = (uint8_t*)dst[(size_t)k * nplanes + p];
  }
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  const uint32_t ntiles = (uint32_t)((n + kPartTile - 1) / kPartTile);
  uint32_t* tile_counts;
  uint8_t* d_remap = nullptr;
  int rc;
  if ((rc = sc.get(&tile_counts, (size_t)world * ntiles))) return rc;
  if (rank_of_bucket) {
    if ((rc = sc.get(&d_remap, 64))) return rc;
    std::memcpy(ctx->mailbox, rank_of_bucket, 64);  // pinned source for the async upload
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_remap, ctx->mailbox, 64, hipMemcpyHostToDevice, ctx->stream));
  }
  {
    PcvProf prof(ctx, PCV_K_PARTITION_COUNT);
    hipLaunchKernelGGL(partition_count_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, n, owner, world, ntiles, tile_counts,
                       (const uint8_t*)d_remap);
  }
  hipLaunchKernelGGL(partition_scan_kernel, dim3(world), dim3(1024), 0, ctx->stream, tile_counts, ntiles);
  {
    PcvProf prof(ctx, PCV_K_PARTITION_SCATTER);
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, n, owner, world, ntiles, tile_counts,
                       pl, (const uint8_t*)d_remap);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

extern "C" uint64_t pcv_route_tiles(uint64_t n) { return (n + kRouteTile - 1) / kRouteTile; }

extern "C" int pcv_route_plan(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, uint8_t* bucket,
                              uint16_t* tile_hist, uint64_t counts[64]) {
  if (!ctx) return PCV_E_INVALID;
  if (!params || !points || !counts) return ctx->fail(PCV_E_INVALID, "null argument");
  for (int b = 0; b < 64; ++b) counts[b] = 0;
  if (points->n == 0) return PCV_OK;  // an empty input slice (a rank without points) is fine
  if (!bucket || !tile_hist) return ctx->fail(PCV_E_INVALID, "bucket / tile_hist is null");
  if (points->mem != PCV_MEM_DEVICE) return ctx->fail(PCV_E_INVALID, "pcv_route_plan works on device-resident points");
  if (points->n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per call");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvLevels lv;
  int max_level;
  pcv_make_levels(params->bbox_min, params->bbox_max, params->resolution, 2, &lv, &max_level, nullptr, nullptr);
  PcvScratch sc(ctx);
  unsigned long long* d_counts;
  int rc;
  if ((rc = sc.get(&d_counts, 64))) return rc;
  PCV_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, 64 * 8, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_ROUTE_BUCKET);
    const bool octants_only = (params->flags & PCV_ROUTE_OCTANTS_ONLY) != 0u && lv.nlevels >= 1;
    // the streaming form wants 16-byte loads and dword stores: aligned bases (pool blocks and torch tensors are; views may not be)
    const bool aligned = ((((uintptr_t)points->x | (uintptr_t)points->y | (uintptr_t)points->z) & 15) | ((uintptr_t)bucket & 3)) == 0;
    if (octants_only && aligned) {
      hipLaunchKernelGGL(route_octants_kernel, dim3((unsigned)pcv_route_tiles(points->n)), dim3(256), 0, ctx->stream, lv, points->n, points->x,
                         points->y, points->z, bucket, tile_hist, d_counts);
      hipLaunchKernelGGL(route_hist_sum_kernel, dim3(kHistSumBlocks), dim3(1024), 0, ctx->stream, tile_hist, (uint32_t)pcv_route_tiles(points->n), d_counts);
    } else
      hipLaunchKernelGGL(route_plan_kernel, dim3((unsigned)pcv_route_tiles(points->n)), dim3(256), 0, ctx->stream, lv, points->n, points->x,
                         points->y, points->z, bucket, tile_hist, d_counts, octants_only);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_counts, 64 * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < 64; ++b) counts[b] = ctx->mailbox[b];
  return PCV_OK;
}

extern "C" int pcv_route_scatter(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, const uint8_t* bucket,
                                 const uint16_t* tile_hist, uint32_t world, const uint8_t rank_of_bucket[64],
                                 const pcv_route_dst* dst) {
  if (!ctx) return PCV_E_INVALID;
  if (!params || !points || !rank_of_bucket || !dst) return ctx->fail(PCV_E_INVALID, "null argument");
  if (world < 1 || world > 8) return ctx->fail(PCV_E_INVALID, "world must be 1..8");
  const uint64_t n = points->n;
  if (n == 0) return PCV_OK;
  if (!bucket || !tile_hist) return ctx->fail(PCV_E_INVALID, "bucket / tile_hist is null");
  if (points->mem != PCV_MEM_DEVICE) return ctx->fail(PCV_E_INVALID, "pcv_route_scatter works on device-resident points");
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per call");
  if (!points->color || (points->color_stride != 3 && points->color_stride != 4))
    return ctx->fail(PCV_E_INVALID, "the level-1 state packs the colour: points->color (stride 3 or 4) is required");
  for (int b = 0; b < 64; ++b)
    if (rank_of_bucket[b] >= world) return ctx->fail(PCV_E_INVALID, "rank_of_bucket entry out of range");
  PcvLevels lv;
  int max_level;
  pcv_make_levels(params->bbox_min, params->bbox_max, params->resolution, 2, &lv, &max_level, nullptr, nullptr);
  if (lv.nlevels < 1 || lv.enc[1] != PCV_ENC_FLOAT32)
    return ctx->fail(PCV_E_INVALID, "the level-1 state is only defined for a Float32-encoded level 1: exchange raw coordinates");
  RouteDst rd{};
  for (uint32_t k = 0; k < world; ++k) {
    rd.orgb[k] = dst[k].oct_rgb, rd.cx[k] = dst[k].cx, rd.cy[k] = dst[k].cy, rd.cz[k] = dst[k].cz, rd.inten[k] = dst[k].intensity;
    // (an owner that receives no row may have null planes)
  }
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  const uint32_t ntiles = (uint32_t)pcv_route_tiles(n);
  uint32_t* tile_counts;
  uint8_t* d_remap;
  int rc;
  if ((rc = sc.get(&tile_counts, (size_t)world * ntiles)) || (rc = sc.get(&d_remap, 64))) return rc;
  std::memcpy(ctx->mailbox, rank_of_bucket, 64);  // pinned source for the async upload
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_remap, ctx->mailbox, 64, hipMemcpyHostToDevice, ctx->stream));
  {
    PcvProf prof(ctx, PCV_K_PARTITION_COUNT);
    hipLaunchKernelGGL(route_owner_counts_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, ctx->stream, tile_hist, ntiles,
                       (const uint8_t*)d_remap, world, tile_counts);
  }
  hipLaunchKernelGGL(partition_scan_kernel, dim3(world), dim3(1024), 0, ctx->stream, tile_counts, ntiles);
  {
    PcvProf prof(ctx, PCV_K_PARTITION_SCATTER);
    hipLaunchKernelGGL(route_scatter_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, lv, n, points->x, points->y, points->z, points->color,
                       points->color_stride, points->intensity, bucket, (const uint8_t*)d_remap, world, ntiles, tile_counts, rd);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

int pcv_launch_aabb(pcv_ctx* ctx, uint64_t n, const double* x, const double* y, const double* z, double* partial,
                    double* out6) {
  uint64_t want = (n + (uint64_t)kAabbBlock * 2 * 8 - 1) / ((uint64_t)kAabbBlock * 2 * 8);
  // two resident workgroups per CU stream best: 512 blocks read 100 M points at 6.3 TB/s, 2048 at 5.5, 256 at 4.6
  static const int maxb = [] {
    if (const char* e = pcv_experiment("PCV_AABB_BLOCKS")) return std::min(2048, std::max(1, atoi(e)));  // experiments
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    return std::min(2048, std::max(64, 2 * cus));
  }();
  int blocks = (int)(want < 1 ? 1 : (want > (uint64_t)maxb ? (uint64_t)maxb : want));
  {
    PcvProf prof(ctx, PCV_K_AABB);
    hipLaunchKernelGGL(aabb_partial_kernel, dim3(blocks), dim3(kAabbBlock), 0, ctx->stream, n, x, y, z, partial);
  }
  hipLaunchKernelGGL(aabb_final_kernel, dim3(1), dim3(256), 0, ctx->stream, blocks, partial, out6);
  return blocks;
}

void pcv_launch_chain_keys(pcv_ctx* ctx, const PcvLevels& lv, uint64_t n, uint64_t stride, const double* x,
                           const double* y, const double* z, void* keys, bool keys32, const PcvRouted& routed,
                           uint32_t clump_shift, uint32_t* zero, size_t zero_words) {
  if (n == 0) return;
  uint4* zv = reinterpret_cast<uint4*>(zero);
  const uint32_t zn = zero ? (uint32_t)((zero_words + 3) / 4) : 0u;
  uint64_t blocks = (n + 255) / 256;
  PcvProf prof(ctx, PCV_K_CHAIN_KEYS);
  if (keys32)
    hipLaunchKernelGGL(chain_keys_kernel<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, lv, n, stride,
                       clump_shift, x, y, z, routed, (uint32_t*)keys, zv, zn);
  else
    hipLaunchKernelGGL(chain_keys_kernel<uint64_t>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, lv, n, stride,
                       clump_shift, x, y, z, routed, (uint64_t*)keys, zv, zn);
}

void pcv_launch_chain_keys_deep(pcv_ctx* ctx, const PcvLevels& lv, uint64_t n, const double* x, const double* y,
                                const double* z, const PcvRouted& routed, uint32_t* const words[4]) {
  if (n == 0) return;
  DeepWords dw{{words[0], words[1], words[2], words[3]}};
  PcvProf prof(ctx, PCV_K_CHAIN_KEYS);
  hipLaunchKernelGGL(chain_keys_deep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lv, n, x, y, z, routed,
                     dw);
}
void pcv_launch_combine_words(pcv_ctx* ctx, uint64_t n, const uint32_t* const words[4], uint64_t* hi, uint64_t* lo) {
  if (n == 0) return;
  DeepWords dw{{(uint32_t*)words[0], (uint32_t*)words[1], (uint32_t*)words[2], (uint32_t*)words[3]}};
  hipLaunchKernelGGL(combine_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, dw, hi, lo);
}

void pcv_launch_depth_probe(pcv_ctx* ctx, const uint64_t* sorted, uint32_t n, uint32_t gap, uint32_t* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(depth_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, sorted, n, gap, out);
}
