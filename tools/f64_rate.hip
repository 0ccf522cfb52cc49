// Micro-benchmark: SIMD cycles a wave64 VALU instruction of each CLASS the chain kernels use occupies on gfx950, and the
// clock the part sustains while it does so. One kernel per class; every lane runs 8 independent chains of ONE instruction
// (inline asm, so the compiler can neither fuse nor drop anything), 8 waves per SIMD, 5 000 x 32 instructions per lane.
//   plain run:            ms per class -> wave-instructions / s
//   under rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace (tools/f64_rate.sh): GUI_ACTIVE / duration = the clock DURING that
//   kernel -> SIMD cycles per wave-instruction = clock x time x 1024 SIMDs / wave-instructions, with no nominal-clock guess
// Build: hipcc -O3 --offload-arch=gfx950 tools/f64_rate.hip -o tools/f64_rate.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>


template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double a, double b) {
  double r[8];
  float f[8];
  uint32_t u[8];
  for (int j = 0; j < 8; ++j) {
    r[j] = a + j + threadIdx.x;
    f[j] = (float)r[j];
    u[j] = threadIdx.x * 8 + j;
  }
  uint32_t ua = (uint32_t)threadIdx.x | 1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
      const int j = jj & 7;
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[j]) : "v"(b));
      if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      if (OP == 3) asm volatile("v_trunc_f64 %0, %0" : "+v"(r[j]));
      if (OP == 4) asm volatile("v_max_f64 %0, %0, %1" : "+v"(r[j]) : "v"(b));
      if (OP == 5) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(r[j]), "v"(b) : "vcc");
      if (OP == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(r[j]));
      if (OP == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[j]) : "v"(f[j]));
      if (OP == 8) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[j]) : "v"(r[j]));
      if (OP == 9) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r[j]) : "v"(u[j]));
      if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(ua));
      if (OP == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[j]) : "v"(ua));
      if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(ua) : );
      if (OP == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[j]) : "v"(ua));
      if (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j]) : "v"(f[(j + 1) & 7]));
      if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(ua));
      if (OP == 16) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[j]), "v"(ua) : "vcc");
    }
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += r[j] + (double)f[j] + (double)u[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

struct Row {
  const char* name;
  const char* cls;
};
static const Row kRows[] = {
    {"v_fma_f64", "f64_arith"},      {"v_add_f64", "f64_arith"},     {"v_mul_f64", "f64_arith"},     {"v_trunc_f64", "f64_other"},
    {"v_max_f64", "f64_other"},      {"v_cmp_gt_f64", "f64_other"},  {"v_cvt_f32_f64", "f64_cvt"},   {"v_cvt_f64_f32", "f64_cvt"},
    {"v_cvt_u32_f64", "f64_cvt"},    {"v_cvt_f64_u32", "f64_cvt"},   {"v_add_u32", "b32"},           {"v_and_b32", "b32"},
    {"v_cndmask_b32", "b32"},        {"v_lshl_or_b32", "b32"},       {"v_fma_f32", "b32"},           {"v_mov_b32", "b32"},
    {"v_cmp_lt_u32", "b32"},
};

template <int OP>
void run(double* d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, 500, 1.0000001, 1e-9);  // warm-up: clocks ramp
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double winst = (double)blocks * 4 * iters * 32;  // wave-instructions of the class (the loop's own SALU is scalar)
  printf("{\"op\": %d, \"inst\": \"%s\", \"class\": \"%s\", \"ms\": %.4f, \"wave_insts\": %.0f, \"cycles_at_2.4GHz\": %.3f}\n", OP, kRows[OP].name,
         kRows[OP].cls, best, winst, best * 1e-3 * 2.4e9 * 1024 / winst);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int blocks = 256 * 8, iters = argc > 1 ? atoi(argv[1]) : 5000;
  double* d;
  hipMalloc(&d, (size_t)blocks * 256 * 8);
  run<0>(d, blocks, iters);
  run<1>(d, blocks, iters);
  run<2>(d, blocks, iters);
  run<3>(d, blocks, iters);
  run<4>(d, blocks, iters);
  run<5>(d, blocks, iters);
  run<6>(d, blocks, iters);
  run<7>(d, blocks, iters);
  run<8>(d, blocks, iters);
  run<9>(d, blocks, iters);
  run<10>(d, blocks, iters);
  run<11>(d, blocks, iters);
  run<12>(d, blocks, iters);
  run<13>(d, blocks, iters);
  run<14>(d, blocks, iters);
  run<15>(d, blocks, iters);
  run<16>(d, blocks, iters);
  hipFree(d);
  return 0;
}
