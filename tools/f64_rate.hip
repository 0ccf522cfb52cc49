// Micro-benchmark: issue rate of f64 VALU ops on gfx950 (how many cycles a wave64 v_fma_f64 / v_add_f64 /
// v_mul_f64 / v_trunc_f64 / v_cvt occupies a SIMD). Build: hipcc -O3 --offload-arch=gfx950 tools/f64_rate.hip -o f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a + j + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) r[j] = __fma_rn(r[j], a, b);
      if (OP == 1) r[j] = r[j] + b;
      if (OP == 2) r[j] = r[j] * a;
      if (OP == 3) r[j] = trunc(r[j] * a);
      if (OP == 4) r[j] = (double)(float)(r[j]) + b;
      if (OP == 5) { float f = __double2float_rn(r[j]); f = fmaf(f, 1.0001f, 0.5f); r[j] = f; }
      if (OP == 6) r[j] = r[j] > b ? r[j] : a;
    }
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += r[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int ops_per_iter) {
  double* d;
  const int blocks = 256 * 8, iters = 20000;
  hipMalloc(&d, blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0000001, 1e-9);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions = blocks * 4 waves * iters * 8 * ops_per_iter ; SIMDs = 1024 ; clock 2.4 GHz
  double winst = (double)blocks * 4 * iters * 8 * ops_per_iter;
  double clk = ms * 1e-3 * 2.4e9 * 1024;
  printf("%-28s %8.3f ms  -> %5.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, ms, clk / winst);
  hipFree(d);
}
int main() {
  run<0>("v_fma_f64", 1);
  run<1>("v_add_f64", 1);
  run<2>("v_mul_f64", 1);
  run<3>("v_mul_f64 + v_trunc_f64", 2);
  run<4>("cvt f64->f32->f64 + add", 3);
  run<5>("cvt + v_fma_f32 + cvt", 3);
  run<6>("v_cmp_f64 + 2 v_cndmask", 3);
  return 0;
}
