// issuebench: issue cost of the VALU instructions the fixed-step cstr kernel is made of, per wave64 instruction, and the
// accuracy of the fp64 reciprocal estimate.  4 waves per SIMD (grid = 4 workgroups of 256 per CU), 8 independent chains per
// lane, exact instructions through inline asm; the shader clock is read from s_memtime against the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/issuebench tools/issuebench.hip && tools/issuebench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define OPS(X) X(0, "v_fma_f64") X(1, "v_add_f64") X(2, "v_mul_f64") X(3, "v_ldexp_f64") X(4, "v_rndne_f64") \
  X(5, "v_cvt_i32_f64") X(6, "v_cndmask_b32") X(7, "v_cmp_lt_f64") X(8, "v_mov_b32") X(9, "v_lshl_add_u32") \
  X(10, "v_rcp_f64") X(11, "v_max_f64") X(12, "v_fma_f32") X(13, "v_pk_fma_f32") X(14, "v_exp_f32") X(15, "v_rcp_f32") \
  X(16, "v_cvt_f32_f64") X(17, "v_cvt_f64_f32") X(18, "v_and_b32") X(19, "v_fmac_f64 (2-operand)") X(20, "v_cmp_class_f64")

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed, long long* clk) {
  double v[8];
  float f[8];
  int n[8];
  for (int i = 0; i < 8; ++i) {
    v[i] = seed + i * 0.125 + threadIdx.x * 1e-3;
    f[i] = (float)v[i];
    n[i] = i + threadIdx.x;
  }
  const double c1 = 0.999999, c2 = 1e-9;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
      if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
      if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
      if (OP == 3) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(n[i]));
      if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i]));
      if (OP == 5) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(v[i]));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
      if (OP == 7) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v[i]), "v"(c1) : "vcc");
      if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) & 7]));
      if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
      if (OP == 10) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i]));
      if (OP == 11) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
      if (OP == 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(0.999f), "v"(1e-6f));
      if (OP == 13) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
      if (OP == 14) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      if (OP == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      if (OP == 16) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(v[i]));
      if (OP == 17) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[i]) : "v"(f[i]));
      if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
      if (OP == 19) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
      if (OP == 20) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(v[i]), "v"(n[0]) : "vcc");
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + f[i] + n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

template <int OP>
void run(const char* name, double fma_ns) {
  const int grid = 256 * 4, iters = 4096;
  double* d;
  long long* c;
  hipMalloc(&d, sizeof(double) * grid * 256);
  hipMalloc(&c, 16);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  k<OP><<<grid, 256>>>(d, 64, 1.5, c);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<grid, 256>>>(d, iters, 1.5, c);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  long long h[2];
  hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
  const double winst = 4.0 * iters * 8;  // wave-instructions per SIMD
  const double ns_per = ms * 1e6 / winst;
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // s_memtime ticks per ns (wall clock: 100 MHz)
  printf("%-26s %8.3f ms  %6.2f ns per wave-instruction  (s_memtime/wall = %.3f GHz)\n", name, ms, ns_per, ghz);
  hipFree(d);
  hipFree(c);
}

__global__ void rcp_acc(const double* x, double* e, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double r = __builtin_amdgcn_rcp(x[i]);
  e[i] = __builtin_fabs(__builtin_fma(-x[i], r, 1.0));  // |1 - x r| = relative error of the estimate (exact residual)
}

int main() {
#define X(id, nm) run<id>(nm, 0);
  OPS(X)
#undef X
  const int n = 1 << 22;
  std::vector<double> x(n), e(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = (i & 1) ? 1.0 + u : 250.0 + 350.0 * u;  // a full binade, and the cstr temperature range
  }
  double *dx, *de;
  hipMalloc(&dx, n * 8), hipMalloc(&de, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  rcp_acc<<<n / 256, 256>>>(dx, de, n);
  hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int i = 0; i < n; ++i) mx = e[i] > mx ? e[i] : mx;
  printf("v_rcp_f64 estimate: max relative error %.3e = 2^%.1f over %d arguments\n", mx, std::log2(mx), n);
  return 0;
}
