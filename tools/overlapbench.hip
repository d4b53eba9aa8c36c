// overlapbench.hip -- does fp64 VALU work hide under the step kernel's HBM traffic?
// Same I/O footprint as the cstr step (3 fp64 rows in, 6 fp64 rows + 1 byte row out, B = 2^20),
// plus a synthetic arithmetic load: CH independent dependent-FMA chains of length L per env.
// Sweeps L for CH = 1,2,4 and a few launch shapes; prints us/launch.
//   hipcc --offload-arch=gfx950 -O3 tools/overlapbench.hip -o tools/overlapbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

struct Args {
  double* x;
  const double* a;
  double* obs;
  double* rew;
  uint8_t* done;
  int64_t B;
  int L;
  double k0, k1;
};

template <int CH>
__device__ __forceinline__ void chains(double (&v)[CH], int L, double k0, double k1) {
  for (int i = 0; i < L; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = __builtin_fma(v[c], k0, k1);
  }
}

// one env per lane, one-shot grid, TB threads per block; the env's work is split in CH chains
template <int CH, int TB>
__global__ __launch_bounds__(TB) void k_one(Args A) {
  const int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x;
  if (e >= A.B) return;
  const double x0 = A.x[e], x1 = A.x[A.B + e], a = A.a[e];
  double v[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) v[c] = x0 + c * a;
  chains<CH>(v, A.L / CH, A.k0, A.k1);
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c];
  A.x[e] = s;
  A.x[A.B + e] = x1 + s;
  A.obs[e] = x0;
  A.obs[A.B + e] = x1;
  A.obs[2 * A.B + e] = a;
  A.rew[e] = s * x1;
  A.done[e] = s > x1;
}

// EPL envs per lane (EPL = 2 or 4): 16-byte accesses, EPL independent chains of full length L
template <int UNR>
__global__ __launch_bounds__(256) void k_vec(Args A) {
  const int64_t base = (int64_t)blockIdx.x * 256 * 2 * UNR + threadIdx.x * 2;
  double2 x0[UNR], x1[UNR], a[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int64_t e = base + (int64_t)u * 512;
    x0[u] = *(const double2*)(A.x + e);
    x1[u] = *(const double2*)(A.x + A.B + e);
    a[u] = *(const double2*)(A.a + e);
  }
  double v[2 * UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    v[2 * u] = x0[u].x;
    v[2 * u + 1] = x0[u].y;
  }
  chains<2 * UNR>(v, A.L, A.k0, A.k1);
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int64_t e = base + (int64_t)u * 512;
    const double2 s = make_double2(v[2 * u], v[2 * u + 1]);
    *(double2*)(A.x + e) = s;
    *(double2*)(A.x + A.B + e) = make_double2(x1[u].x + s.x, x1[u].y + s.y);
    *(double2*)(A.obs + e) = x0[u];
    *(double2*)(A.obs + A.B + e) = x1[u];
    *(double2*)(A.obs + 2 * A.B + e) = a[u];
    *(double2*)(A.rew + e) = make_double2(s.x * x1[u].x, s.y * x1[u].y);
    *(uint16_t*)(A.done + e) = (uint16_t)((s.x > x1[u].x) | ((s.y > x1[u].y) << 8));
  }
}

int main() {
  const int64_t B = 1 << 20;
  const int NA = 64;
  double *x, *a, *obs, *rew;
  uint8_t* done;
  CK(hipMalloc(&x, 2 * B * 8));
  CK(hipMalloc(&a, (size_t)NA * B * 8));
  CK(hipMalloc(&obs, 3 * B * 8));
  CK(hipMalloc(&rew, B * 8));
  CK(hipMalloc(&done, B));
  CK(hipMemset(x, 0, 2 * B * 8));
  CK(hipMemset(a, 0, (size_t)NA * B * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](auto launch) -> float {
    for (int i = 0; i < 10; ++i) launch(i);
    hipDeviceSynchronize();
    const int n = 200;
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) launch(i);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / n * 1e3f;
  };
  const int Ls[] = {0, 64, 128, 256, 512, 1024};
  printf("us/launch; L = dependent fp64 FMAs per env (x-axis), 2^20 envs, I/O footprint 73 B/env\n");
  printf("%-36s", "shape \\ L");
  for (int L : Ls) printf("%8d", L);
  printf("\n");
  auto row = [&](const char* name, auto mk) {
    printf("%-36s", name);
    for (int L : Ls) {
      float us = run([&](int i) { mk(Args{x, a + (size_t)(i % NA) * B, obs, rew, done, B, L, 0.999999, 1e-9}); });
      printf("%8.2f", us);
    }
    printf("\n");
  };
  row("1 env/lane CH=1 TB=256", [&](Args g) { hipLaunchKernelGGL((k_one<1, 256>), dim3(B / 256), dim3(256), 0, 0, g); });
  row("1 env/lane CH=2 TB=256", [&](Args g) { hipLaunchKernelGGL((k_one<2, 256>), dim3(B / 256), dim3(256), 0, 0, g); });
  row("1 env/lane CH=4 TB=256", [&](Args g) { hipLaunchKernelGGL((k_one<4, 256>), dim3(B / 256), dim3(256), 0, 0, g); });
  row("1 env/lane CH=1 TB=64", [&](Args g) { hipLaunchKernelGGL((k_one<1, 64>), dim3(B / 64), dim3(64), 0, 0, g); });
  row("1 env/lane CH=4 TB=64", [&](Args g) { hipLaunchKernelGGL((k_one<4, 64>), dim3(B / 64), dim3(64), 0, 0, g); });
  row("2 env/lane (2 chains) TB=256", [&](Args g) { hipLaunchKernelGGL((k_vec<1>), dim3(B / 512), dim3(256), 0, 0, g); });
  row("4 env/lane (4 chains) TB=256", [&](Args g) { hipLaunchKernelGGL((k_vec<2>), dim3(B / 1024), dim3(256), 0, 0, g); });
  row("8 env/lane (8 chains) TB=256", [&](Args g) { hipLaunchKernelGGL((k_vec<4>), dim3(B / 2048), dim3(256), 0, 0, g); });
  printf("VALU-only time at fp64 FMA peak (4 cyc/wave-instr, 1024 SIMDs, 2.4 GHz): L=256 -> %.2f us, L=1024 -> %.2f us\n",
         256.0 * 16384 / 1024 * 4 / 2400.0, 1024.0 * 16384 / 1024 * 4 / 2400.0);
  return 0;
}
