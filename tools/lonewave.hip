// lonewave: what does ONE wave per SIMD pay per v_fma_f64, by the number of independent chains it carries -- and what do scalar
// moves interleaved with them cost?  (the work-queue kernels run one wave per SIMD; the default cstr plan's fix-up launch is
// one lane's chain).  Grid = waves_per_simd x 4 x CUs single-wave workgroups; s_memtime / wall clock gives the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lonewave tools/lonewave.hip && /tmp/lonewave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH, int SMOV>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed, long long* clk) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
  const double c1 = 0.999999, c2 = 1e-9;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (SMOV >= 1) asm volatile("s_mov_b32 s40, 0x3ff00000" ::: "s40");
        if (SMOV >= 2) asm volatile("s_mov_b32 s41, 0x3ff00001" ::: "s41");
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int CH, int SMOV>
void run(int wps, int cus, double* out, long long* clk) {
  const int iters = 20000, grid = wps * 4 * cus;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CH, SMOV>), dim3(grid), dim3(64), 0, 0, out, 100, 1.0, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CH, SMOV>), dim3(grid), dim3(64), 0, 0, out, iters, 1.0, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, clk, sizeof c, hipMemcpyDeviceToHost);
  const double per_fma_ns = ms * 1e6 / ((double)iters * 8);  // per wave-instruction of ONE wave (waves run concurrently)
  printf("waves/SIMD %d  chains %d  s_mov per fma %d : %7.2f ns per v_fma_f64 of a wave = %5.2f cycles at the kernel's own clock (%lld ticks of s_memtime)\n",
         wps, CH, SMOV, per_fma_ns, per_fma_ns * (double)c / (ms * 1e6) * 1.0, c);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  long long* clk;
  hipMalloc(&out, sizeof(double) * 64 * 8 * 4 * cus);
  hipMalloc(&clk, 64);
  for (int wps : {1, 2, 4}) {
    run<1, 0>(wps, cus, out, clk);
    run<2, 0>(wps, cus, out, clk);
    run<4, 0>(wps, cus, out, clk);
    run<8, 0>(wps, cus, out, clk);
    run<1, 1>(wps, cus, out, clk);
    run<1, 2>(wps, cus, out, clk);
    run<8, 1>(wps, cus, out, clk);
    run<8, 2>(wps, cus, out, clk);
  }
  return 0;
}
