// What do cross-stream dependencies cost the PRODUCING stream? (gfx950, ROCm 7)
//   a) k1 -> k2 back to back                       b) k1 -> hipEventRecord -> k2 (the fork of a side stream)
//   c) k1 -> k2 where k2's first lane stores a flag that a side stream waits for with hipStreamWaitValue32
// prints the time from k1's start to k2's end (HIP events around the pair, 200 repetitions, median) and whether the side stream's
// kernel really ran after the flag.   build: hipcc --offload-arch=gfx950 -O2 tools/stream_ops_probe.hip -o tools/stream_ops_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ void busy(uint32_t* out, int iters) {
  uint32_t v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
  if (v == 0x12345u) out[0] = v;
}
__global__ void busy_flag(uint32_t* out, int iters, uint32_t* flag, uint32_t seq) {
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint32_t v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
  if (v == 0x12345u) out[0] = v;
}
__global__ void observe(const uint32_t* flag, uint32_t* seen) { seen[0] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
int main() {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  uint32_t *out, *seen, *flag = nullptr;
  CK(hipMalloc(&out, 64));
  CK(hipHostMalloc(&seen, 64));
  hipError_t fe = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(signal): %s\n", hipGetErrorString(fe));
  if (fe != hipSuccess) CK(hipHostMalloc(&flag, 8));
  CK(hipMemset(flag, 0, 8));
  hipEvent_t e0, e1, fork;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  const int reps = 200, iters = 2000;
  hipEvent_t join;
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  for (int mode = 0; mode < 7; ++mode) {
    std::vector<float> ms;
    int wrong = 0;
    for (int r = 0; r < reps; ++r) {
      const uint32_t seq = (uint32_t)(mode * 1000 + r + 1);
      seen[0] = 0;
      CK(hipEventRecord(e0, a));
      hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, a, out, iters);
      if (mode == 1) { CK(hipEventRecord(fork, a)); CK(hipStreamWaitEvent(b, fork, 0)); hipLaunchKernelGGL(observe, dim3(1), dim3(1), 0, b, flag, seen); }
      if (mode == 2) { hipError_t w = hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueEq, 0xffffffffu); if (w != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(w)); return 0; }
                       hipLaunchKernelGGL(observe, dim3(1), dim3(1), 0, b, flag, seen); }
      // joins: the side stream does a little work that ended long ago; `a` waits for it between k1 and k2
      if (mode == 4) { hipLaunchKernelGGL(observe, dim3(1), dim3(1), 0, b, flag, seen); CK(hipEventRecord(join, b)); CK(hipStreamWaitEvent(a, join, 0)); }
      if (mode == 5) { hipLaunchKernelGGL(observe, dim3(1), dim3(1), 0, b, flag, seen); CK(hipStreamWriteValue32(b, flag, seq, 0)); CK(hipStreamWaitValue32(a, flag, seq, hipStreamWaitValueGte, 0xffffffffu)); }
      if (mode == 6) { CK(hipStreamWaitValue32(a, flag, 0, hipStreamWaitValueGte, 0xffffffffu)); }  // a wait that is already satisfied
      if (mode == 2 || mode == 3) hipLaunchKernelGGL(busy_flag, dim3(64), dim3(256), 0, a, out, iters, flag, seq);
      else hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, a, out, iters);
      CK(hipEventRecord(e1, a));
      CK(hipEventSynchronize(e1));
      CK(hipStreamSynchronize(b));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      ms.push_back(t);
      if (mode == 2 && seen[0] != seq) ++wrong;
    }
    std::sort(ms.begin(), ms.end());
    const char* names[7] = {"k1 -> k2", "k1 -> eventRecord(fork) -> k2", "k1 -> k2(stores flag), side: waitValue32 -> observe", "k1 -> k2(stores flag), no side stream",
                            "k1 -> waitEvent(side's event) -> k2", "k1 -> waitValue32(flag the side stream wrote) -> k2", "k1 -> waitValue32(already true) -> k2"};
    printf("%-56s median %.1f us  min %.1f us  side saw a stale flag: %d of %d\n", names[mode], ms[reps / 2] * 1e3, ms[0] * 1e3, wrong, reps);
  }
  return 0;
}
