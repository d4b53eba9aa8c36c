// transbench: issue cost of the fp64 transcendental estimates against a fused multiply-add, per wave64 instruction.
// One workgroup of 256 threads per CU x 4 resident, 8 independent chains per lane, 4096 iterations.
//   hipcc --offload-arch=gfx950 -O3 -o tools/transbench tools/transbench.hip && tools/transbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) v[i] = __builtin_fma(v[i], 0.999999, 1e-9);
      if (OP == 1) v[i] = __builtin_amdgcn_rsq(v[i]);
      if (OP == 2) v[i] = __builtin_amdgcn_rcp(v[i]);
      if (OP == 3) v[i] = __builtin_amdgcn_sqrt(v[i]);
      if (OP == 4) v[i] = (double)__builtin_amdgcn_rsqf((float)v[i]);
      if (OP == 5) v[i] = (double)__builtin_amdgcn_rcpf((float)v[i]);
      if (OP == 6) { float f = (float)v[i]; f = __builtin_amdgcn_rsqf(f); v[i] = (double)f; v[i] = __builtin_fma(v[i], 0.999999, 1e-9); }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, int ops_per_iter) {
  const int grid = 256 * 4, iters = 4096;
  double* d;
  hipMalloc(&d, sizeof(double) * grid * 256);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  k<OP><<<grid, 256>>>(d, 64, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<grid, 256>>>(d, iters, 1.5);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  // wave-instructions per SIMD: 4 waves/SIMD x iters x 8 x ops
  const double winst = 4.0 * iters * 8 * ops_per_iter;
  const double ns_per = ms * 1e6 / winst;
  printf("%-34s %8.3f ms  %6.2f ns per wave-instruction-group (~%.1f cycles at 2.1 GHz)\n", name, ms, ns_per, ns_per * 2.1);
  hipFree(d);
  return ns_per;
}

int main() {
  run<0>("v_fma_f64", 1);
  run<1>("v_rsq_f64", 1);
  run<2>("v_rcp_f64", 1);
  run<3>("v_sqrt_f64", 1);
  run<4>("cvt f64->f32, v_rsq_f32, cvt f32->f64", 1);
  run<5>("cvt f64->f32, v_rcp_f32, cvt f32->f64", 1);
  run<6>("cvt, v_rsq_f32, cvt, v_fma_f64", 1);
  return 0;
}
