// Feasibility probe (not part of the library): how fast is ONE scatter pass that sends 16-byte records straight to
// their final slot (6 073 leaves, stable order), compared with a straight copy? Decides whether a one-pass counting
// sort by leaf could replace the two LDS-staged radix passes of the record sort.
//   hipcc -O3 --offload-arch=gfx950 tools/scatter_probe.hip -o /tmp/scatter_probe && /tmp/scatter_probe [n] [leaves]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const uint32_t leaves = argc > 2 ? (uint32_t)atoi(argv[2]) : 6073u;
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
