// Feasibility probes for the record sort (not part of the library).
//
//   scatter_probe one [n] [leaves]   round 2: ONE scatter pass sending 16-byte records straight to their final slot vs a copy
//   scatter_probe runs [n]           round 4 (VERDICT r03 #4a): the HBM ceiling of the record downsweep's MEMORY PATTERN — 12-byte
//                                    records (u32 key + uint2 payload, two arrays) read in tiles of 8 192 by G workgroups of 1 024
//                                    lanes and written as runs of R = 8 192 / D records into D x G regions ordered (digit,
//                                    workgroup) — with no ranking, no LDS and no barrier at all: registers in, registers out.
//                                    Whatever the real kernel (downsweep_rec12_kernel) does on top can only be slower than this.
//                                    Variants: run length 32 / 64 / 128 / 256 records, region bases aligned to 256 B or skewed by
//                                    an odd number of records (the real sort's bases are arbitrary), workgroup -> piece mapping
//                                    plain or XCD-major (consecutive pieces on one XCD: the partial lines where two
//                                    neighbouring regions meet then merge in ONE L2 instead of two).
//   hipcc -O3 --offload-arch=gfx950 tools/scatter_probe.hip -o tools/scatter_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ __launch_bounds__(256) void scatter_kernel(uint64_t n, const uint32_t* __restrict__ dest,
                                                       const uint4* __restrict__ in, uint4* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[dest[i]] = in[i];
}
__global__ __launch_bounds__(256) void copy_kernel(uint64_t n, const uint32_t* __restrict__ dest, const uint4* __restrict__ in,
                                                    uint4* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// One workgroup per piece of `chunk` records (a multiple of 8 192). Tile t of piece g: record p of the tile (p = j * 1024 + lane,
// as the real kernel's output phase walks it) belongs to digit d = p / R and goes to
//   out[base(d, g) + t * R + p % R],  base(d, g) = d * stride_d + g * stride_g + skew(d, g)
// so every digit's region is appended to by R records per tile, exactly the real kernel's write pattern with perfectly
// balanced digits. The next tile's loads are issued before this tile's stores (as the real kernel prefetches).
template <int LOG_R, bool XCD>
__global__ __launch_bounds__(1024) void runs_kernel(const uint32_t* __restrict__ kin, const uint2* __restrict__ vin, uint32_t* __restrict__ kout,
                                                    uint2* __restrict__ vout, uint32_t chunk, uint32_t groups, uint32_t stride_d,
                                                    uint32_t stride_g, uint32_t skew_on) {
  constexpr uint32_t R = 1u << LOG_R;
  const uint32_t b = blockIdx.x;
  const uint32_t g = XCD ? (b & 7u) * (groups >> 3) + (b >> 3) : b;
  const uint32_t t = threadIdx.x;
  const uint64_t in0 = (uint64_t)g * chunk;
  uint32_t k[8];
  uint2 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    k[j] = kin[in0 + j * 1024 + t];
    v[j] = vin[in0 + j * 1024 + t];
  }
  const uint32_t tiles = chunk >> 13;
  for (uint32_t tile = 0; tile < tiles; ++tile) {
    uint32_t k2[8];
    uint2 v2[8];
    if (tile + 1 < tiles) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        k2[j] = kin[in0 + (uint64_t)(tile + 1) * 8192 + j * 1024 + t];
        v2[j] = vin[in0 + (uint64_t)(tile + 1) * 8192 + j * 1024 + t];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t p = j * 1024 + t, d = p >> LOG_R;
      const uint32_t skew = skew_on ? ((d * 131u + g * 17u) & 31u) | 1u : 0u;
      const uint64_t o = (uint64_t)d * stride_d + (uint64_t)g * stride_g + skew + tile * R + (p & (R - 1u));
      kout[o] = k[j];
      vout[o] = v[j];
    }
    if (tile + 1 < tiles) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        k[j] = k2[j];
        v[j] = v2[j];
      }
    }
  }
}
// round 6: the same pattern with the record as ONE 12-byte element (key and payload side by side): a run of R records is one
// contiguous piece of 12 R bytes instead of two pieces of 4 R and 8 R bytes — half as many write streams, half as many partial
// lines where runs begin and end
struct Rec12 {
  uint32_t a, b, c;
};
template <int LOG_R>
__global__ __launch_bounds__(1024) void runs_aos_kernel(const Rec12* __restrict__ in, Rec12* __restrict__ out, uint32_t chunk, uint32_t groups,
                                                        uint32_t stride_d, uint32_t stride_g, uint32_t skew_on) {
  constexpr uint32_t R = 1u << LOG_R;
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  const uint64_t in0 = (uint64_t)g * chunk;
  Rec12 r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = in[in0 + j * 1024 + t];
  const uint32_t tiles = chunk >> 13;
  for (uint32_t tile = 0; tile < tiles; ++tile) {
    Rec12 r2[8];
    if (tile + 1 < tiles) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r2[j] = in[in0 + (uint64_t)(tile + 1) * 8192 + j * 1024 + t];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t p = j * 1024 + t, d = p >> LOG_R;
      const uint32_t skew = skew_on ? ((d * 131u + g * 17u) & 31u) | 1u : 0u;
      out[(uint64_t)d * stride_d + (uint64_t)g * stride_g + skew + tile * R + (p & (R - 1u))] = r[j];
    }
    if (tile + 1 < tiles) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = r2[j];
    }
  }
}
template <int LOG_R>
static int time_runs_aos(const void* in, void* out, uint32_t chunk, uint32_t groups, uint32_t skew, hipEvent_t a, hipEvent_t b, double bytes) {
  const uint32_t D = 8192u >> LOG_R;
  const uint32_t per = chunk / D + 64;
  const uint32_t stride_g = (per + 63u) & ~63u;
  const uint32_t stride_d = stride_g * groups;
  float best = 1e9f, sum = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((runs_aos_kernel<LOG_R>), dim3(groups), dim3(1024), 0, 0, (const Rec12*)in, (Rec12*)out, chunk, groups, stride_d, stride_g, skew);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep) {
      best = std::min(best, ms);
      sum += ms;
    }
  }
  printf("{\"pattern\": \"runs, 12-byte records as ONE array\", \"run_records\": %u, \"digit_values\": %u, \"bases\": \"%s\", \"best_ms\": %.4f, "
         "\"mean_ms\": %.4f, \"GBps_best\": %.1f, \"frac_of_8TBps\": %.3f}\n",
         1u << LOG_R, D, skew ? "skewed (odd record offsets)" : "aligned", best, sum / 5, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12);
  fflush(stdout);
  return 0;
}

__global__ __launch_bounds__(1024) void copy12_kernel(const uint32_t* __restrict__ kin, const uint2* __restrict__ vin, uint32_t* __restrict__ kout,
                                                      uint2* __restrict__ vout, uint32_t chunk) {
  const uint64_t in0 = (uint64_t)blockIdx.x * chunk;
  for (uint32_t i = threadIdx.x; i < chunk; i += 1024) {
    kout[in0 + i] = kin[in0 + i];
    vout[in0 + i] = vin[in0 + i];
  }
}

static int run_one(uint64_t n, uint32_t leaves) {
  std::vector<uint32_t> rank(n), dest(n), count(leaves + 1, 0);
  std::mt19937_64 rng(1);
  for (uint64_t i = 0; i < n; ++i) {
    rank[i] = (uint32_t)(rng() % leaves);
    ++count[rank[i] + 1];
  }
  std::partial_sum(count.begin(), count.end(), count.begin());
  for (uint64_t i = 0; i < n; ++i) dest[i] = count[rank[i]]++;  // stable counting sort destination
  uint32_t* d_dest;
  uint4 *d_in, *d_out;
  CK(hipMalloc(&d_dest, n * 4));
  CK(hipMalloc(&d_in, n * 16));
  CK(hipMalloc(&d_out, n * 16));
  CK(hipMemcpy(d_dest, dest.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_in, 1, n * 16));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a));
      if (which == 0) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, n, d_dest, d_in, d_out);
      else hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, 0, n, d_dest, d_in, d_out);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      best = std::min(best, ms);
    }
    printf("%s: %.3f ms for %llu records of 16 B (%u leaves)\n", which ? "scatter to final slots" : "straight copy", best,
           (unsigned long long)n, leaves);
  }
  return 0;
}

template <int LOG_R, bool XCD>
static int time_runs(const uint32_t* kin, const uint2* vin, uint32_t* kout, uint2* vout, uint32_t chunk, uint32_t groups, uint32_t skew,
                     hipEvent_t a, hipEvent_t b, double bytes) {
  const uint32_t D = 8192u >> LOG_R;
  const uint32_t per = chunk / D + 64;                // room of one (digit, workgroup) region, records
  const uint32_t stride_g = (per + 63u) & ~63u;       // 256-byte aligned key regions when not skewed
  const uint32_t stride_d = stride_g * groups;
  float best = 1e9f, sum = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((runs_kernel<LOG_R, XCD>), dim3(groups), dim3(1024), 0, 0, kin, vin, kout, vout, chunk, groups, stride_d, stride_g, skew);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep) {
      best = std::min(best, ms);
      sum += ms;
    }
  }
  printf("{\"pattern\": \"runs\", \"run_records\": %u, \"digit_values\": %u, \"bases\": \"%s\", \"mapping\": \"%s\", \"best_ms\": %.4f, \"mean_ms\": %.4f, "
         "\"GBps_best\": %.1f, \"frac_of_8TBps\": %.3f}\n",
         1u << LOG_R, D, skew ? "skewed (odd record offsets)" : "256-byte aligned", XCD ? "xcd-major" : "plain", best, sum / 5, bytes / (best * 1e-3) / 1e9,
         bytes / (best * 1e-3) / 8e12);
  fflush(stdout);
  return 0;
}

static int run_runs(uint64_t n_req) {
  const uint32_t groups = 1024;
  const uint32_t chunk = (uint32_t)((n_req / groups) & ~8191ull);  // whole tiles
  const uint64_t n = (uint64_t)chunk * groups;
  const uint64_t room = n + (uint64_t)groups * 8192u / 32u * 128u + (1u << 20);  // regions are padded: the smallest run has 256 digit values
  uint32_t *kin, *kout;
  uint2 *vin, *vout;
  CK(hipMalloc(&kin, n * 4));
  CK(hipMalloc(&vin, n * 8));
  CK(hipMalloc(&kout, room * 4));
  CK(hipMalloc(&vout, room * 8));
  CK(hipMemset(kin, 1, n * 4));
  CK(hipMemset(vin, 2, n * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const double bytes = 24.0 * (double)n;
  printf("{\"records\": %llu, \"record_bytes\": 12, \"workgroups\": %u, \"lanes\": 1024, \"tile\": 8192, \"bytes_moved\": %.0f}\n", (unsigned long long)n,
         groups, bytes);
  {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(copy12_kernel, dim3(groups), dim3(1024), 0, 0, kin, vin, kout, vout, chunk);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (rep) best = std::min(best, ms);
    }
    printf("{\"pattern\": \"straight copy (same arrays, same workgroups)\", \"best_ms\": %.4f, \"GBps_best\": %.1f, \"frac_of_8TBps\": %.3f}\n", best,
           bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12);
  }
  {  // round 6: the AoS form, in buffers of its own (12 n bytes each way, the same padding)
    void *ain, *aout;
    CK(hipMalloc(&ain, n * 12));
    CK(hipMalloc(&aout, room * 12));
    CK(hipMemset(ain, 3, n * 12));
    for (uint32_t skew = 0; skew < 2; ++skew) {
      if (time_runs_aos<5>(ain, aout, chunk, groups, skew, a, b, bytes)) return 1;
      if (time_runs_aos<6>(ain, aout, chunk, groups, skew, a, b, bytes)) return 1;
      if (time_runs_aos<7>(ain, aout, chunk, groups, skew, a, b, bytes)) return 1;
    }
    CK(hipFree(ain));
    CK(hipFree(aout));
  }
  for (uint32_t skew = 0; skew < 2; ++skew) {
    if (time_runs<5, false>(kin, vin, kout, vout, chunk, groups, skew, a, b, bytes)) return 1;
    if (time_runs<6, false>(kin, vin, kout, vout, chunk, groups, skew, a, b, bytes)) return 1;
    if (time_runs<7, false>(kin, vin, kout, vout, chunk, groups, skew, a, b, bytes)) return 1;
    if (time_runs<8, false>(kin, vin, kout, vout, chunk, groups, skew, a, b, bytes)) return 1;
    if (time_runs<6, true>(kin, vin, kout, vout, chunk, groups, skew, a, b, bytes)) return 1;
    if (time_runs<7, true>(kin, vin, kout, vout, chunk, groups, skew, a, b, bytes)) return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  const bool runs = argc > 1 && !strcmp(argv[1], "runs");
  const int a0 = (argc > 1 && (runs || !strcmp(argv[1], "one"))) ? 2 : 1;
  const uint64_t n = argc > a0 ? strtoull(argv[a0], nullptr, 10) : 100000000ull;
  if (runs) return run_runs(n);
  const uint32_t leaves = argc > a0 + 1 ? (uint32_t)atoi(argv[a0 + 1]) : 6073u;
  return run_one(n, leaves);
}
