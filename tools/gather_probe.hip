// Calibration of the FETCH_SIZE / WRITE_SIZE counters on the settle pass's access pattern (not part of the library).
// MI355X_MICROARCH.md calibrates "HBM bytes = 2 x FETCH_SIZE" on wide coalesced reads; promote_settle_leaf_kernel mixes a
// coalesced 12-byte record stream with 16-byte gathers from the dense `wide` pool (100 MB at 100 M points), and the counters
// said 1.5 x its algorithmic bytes for three rounds without anyone knowing what a 16-byte gather costs in them.
// Kernels, each launched once over KNOWN byte counts (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; tools/gather_probe.sh):
//   stream_kernel       n records: u32 key + uint2 payload in (12 B), 8 B out                    -> calibrates the coalesced part
//   gather_kernel<S>    m gathers of 16 B, every pool entry exactly once, consecutive lanes S entries apart in a permuted
//                       order (S = 1: neighbouring lanes share lines; S large: every gather is a line of its own), 4 B out
//   settle_like_kernel  the mix: n records streamed, a fraction f of them gathers its pool entry (entry order scrambled
//                       inside slices of 1 024 like the pool's per-wave reservations), 8 B out
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/gather_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ __launch_bounds__(256) void stream_kernel(uint64_t n, const uint32_t* __restrict__ key, const uint2* __restrict__ vec,
                                                      uint2* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const uint2 v = vec[i];
    out[i] = make_uint2(v.x ^ key[i], v.y);
  }
}

// entry visited by gather j: a bijection of [0, m) (m a power of two): bit-reversed low bits spread neighbouring lanes
template <int SPREAD_BITS>
__device__ __forceinline__ uint32_t entry_of(uint32_t j, uint32_t mbits) {
  if (SPREAD_BITS == 0) return j;
  // rotate the index left by SPREAD_BITS inside mbits bits: lanes j, j + 1 land 2^SPREAD_BITS entries apart
  const uint32_t mask = (1u << mbits) - 1u;
  return ((j << SPREAD_BITS) | (j >> (mbits - SPREAD_BITS))) & mask;
}
template <int SPREAD_BITS>
__global__ __launch_bounds__(256) void gather_kernel(uint32_t mbits, const uint4* __restrict__ pool, uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  const uint4 e = pool[entry_of<SPREAD_BITS>(j, mbits)];
  out[j] = e.x ^ e.y ^ e.z ^ e.w;
}

__global__ __launch_bounds__(256) void settle_like_kernel(uint64_t n, const uint32_t* __restrict__ key, const uint2* __restrict__ vec,
                                                           const uint4* __restrict__ pool, uint32_t every, uint2* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint2 v = vec[i];
  const uint32_t k = key[i];
  if (i % every == 0) {  // this record's codes wait in the pool: entry i / every, scrambled inside its slice of 1 024 entries
    const uint32_t e = (uint32_t)(i / every);
    const uint32_t s = (e & ~1023u) | (((e & 1023u) * 389u + 17u) & 1023u);
    const uint4 w = pool[s];
    v.x ^= w.x ^ w.z;
    v.y ^= w.y;
  }
  out[i] = make_uint2(v.x ^ k, v.y);
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const uint32_t mbits = 23;  // 8 388 608 pool entries = 128 MiB
  const uint32_t m = 1u << mbits;
  const uint32_t every = 16;  // one record in 16 gathers (6.25 %; the bench cloud: 6.3 %)
  uint32_t *key, *gout;
  uint2 *vec, *out;
  uint4* pool;
  CK(hipMalloc(&key, n * 4));
  CK(hipMalloc(&vec, n * 8));
  CK(hipMalloc(&out, n * 8));
  CK(hipMalloc(&pool, (size_t)m * 16));
  CK(hipMalloc(&gout, (size_t)m * 4));
  CK(hipMemset(key, 1, n * 4));
  CK(hipMemset(vec, 2, n * 8));
  CK(hipMemset(pool, 3, (size_t)m * 16));
  const unsigned gn = (unsigned)((n + 255) / 256), gm = m / 256;
  for (int rep = 0; rep < 2; ++rep) {  // (the second launch of each is the one to read: same bytes, warm TLBs)
    hipLaunchKernelGGL(stream_kernel, dim3(gn), dim3(256), 0, 0, n, key, vec, out);
    hipLaunchKernelGGL(gather_kernel<0>, dim3(gm), dim3(256), 0, 0, mbits, pool, gout);
    hipLaunchKernelGGL(gather_kernel<3>, dim3(gm), dim3(256), 0, 0, mbits, pool, gout);
    hipLaunchKernelGGL(gather_kernel<10>, dim3(gm), dim3(256), 0, 0, mbits, pool, gout);
    hipLaunchKernelGGL(settle_like_kernel, dim3(gn), dim3(256), 0, 0, n, key, vec, pool, every, out);
  }
  CK(hipDeviceSynchronize());
  printf("{\"records\": %llu, \"pool_entries\": %u, \"gather_every\": %u, \"algorithmic_bytes\": {\"stream_kernel\": {\"read\": %llu, \"write\": %llu}, "
         "\"gather_kernel\": {\"read\": %llu, \"write\": %llu}, \"settle_like_kernel\": {\"read\": %llu, \"write\": %llu}}}\n",
         (unsigned long long)n, m, every, (unsigned long long)(n * 12), (unsigned long long)(n * 8), (unsigned long long)m * 16,
         (unsigned long long)m * 4, (unsigned long long)(n * 12 + (n / every) * 16), (unsigned long long)(n * 8));
  return 0;
}
