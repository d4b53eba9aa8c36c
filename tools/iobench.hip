// iobench.hip -- I/O-pattern microbenchmark for the cstr step kernel's memory footprint
// (3 read rows, 6 fp64 write rows + 1 byte row per env; B = 2^20), no arithmetic.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/iobench.hip -o /tmp/iobench && /tmp/iobench
// Used to find which access shape reaches the memory-system ceiling before changing the product kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

struct Args {
  const double* x;  // [2][B]
  const double* a;  // [1][B]
  double* xo;       // [2][B]  (in place in the product; separate here is equivalent traffic)
  double* obs;      // [3][B]
  double* rew;      // [B]
  uint8_t* done;    // [B]
  int64_t B;
};

// K1: one env per lane, 8-byte accesses, one-shot grid
__global__ __launch_bounds__(256) void k1(Args A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= A.B) return;
  const double x0 = A.x[e], x1 = A.x[A.B + e], a = A.a[e];
  A.xo[e] = x0 + a;
  A.xo[A.B + e] = x1 + a;
  A.obs[e] = x0;
  A.obs[A.B + e] = x1;
  A.obs[2 * A.B + e] = a;
  A.rew[e] = x0 * x1;
  A.done[e] = x0 > x1;
}

template <bool DONE, bool NT>
__device__ __forceinline__ void body2(const Args& A, int64_t e) {
  const double2 x0 = *(const double2*)(A.x + e), x1 = *(const double2*)(A.x + A.B + e), a = *(const double2*)(A.a + e);
  double2 o0 = make_double2(x0.x + a.x, x0.y + a.y), o1 = make_double2(x1.x + a.x, x1.y + a.y);
  double2 r = make_double2(x0.x * x1.x, x0.y * x1.y);
  if (NT) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    auto nt = [](const double2& v, double* p) { __builtin_nontemporal_store(d2{v.x, v.y}, (d2*)p); };
    nt(o0, A.xo + e);
    nt(o1, A.xo + A.B + e);
    nt(x0, A.obs + e);
    nt(x1, A.obs + A.B + e);
    nt(a, A.obs + 2 * A.B + e);
    nt(r, A.rew + e);
  } else {
    *(double2*)(A.xo + e) = o0;
    *(double2*)(A.xo + A.B + e) = o1;
    *(double2*)(A.obs + e) = x0;
    *(double2*)(A.obs + A.B + e) = x1;
    *(double2*)(A.obs + 2 * A.B + e) = a;
    *(double2*)(A.rew + e) = r;
  }
  if (DONE) *(uint16_t*)(A.done + e) = (uint16_t)((x0.x > x1.x) | ((x0.y > x1.y) << 8));
}

// K2: two envs per lane (16-byte accesses), one-shot grid
template <bool DONE, bool NT>
__global__ __launch_bounds__(256) void k2(Args A) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= A.B) return;
  body2<DONE, NT>(A, e);
}

// K4: four envs per lane (2 x 16 B per row), one-shot grid
__global__ __launch_bounds__(256) void k4(Args A) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t half = A.B / 2;
  if (e >= half) return;
  body2<true, false>(A, e);
  body2<true, false>(A, e + half);
}

// K5: persistent grid-stride, two envs per lane
template <bool DONE>
__global__ __launch_bounds__(256) void k5(Args A) {
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; e < A.B; e += (int64_t)gridDim.x * 512)
    body2<DONE, false>(A, e);
}

// K6: done flags packed by a ballot: one 8-byte store per wave instead of 64 one/two-byte stores
__global__ __launch_bounds__(256) void k6(Args A) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= A.B) return;
  const double x0 = A.x[e], x1 = A.x[A.B + e], a = A.a[e];
  A.xo[e] = x0 + a;
  A.xo[A.B + e] = x1 + a;
  A.obs[e] = x0;
  A.obs[A.B + e] = x1;
  A.obs[2 * A.B + e] = a;
  A.rew[e] = x0 * x1;
  const unsigned long long m = __ballot(x0 > x1);
  if ((threadIdx.x & 63) == 0) ((unsigned long long*)A.done)[e >> 6] = m;  // bitmask variant (1 bit/env)
}

// K2P: K2 with non-temporal stores and a padded component stride `ld` (instead of B) -- probes whether the seven
// streams of the step (x0, x1, a, obs0..2, rew), 8 MiB apart in the product layout, alias in the DRAM channel / bank map
struct ArgsP {
  const double* x;
  const double* a;
  double* xo;
  double* obs;
  double* rew;
  uint8_t* done;
  int64_t B, ld;
};
__global__ __launch_bounds__(256) void k2p(ArgsP A) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= A.B) return;
  const d2 x0 = *(const d2*)(A.x + e), x1 = *(const d2*)(A.x + A.ld + e), a = *(const d2*)(A.a + e);
  __builtin_nontemporal_store(x0 + a, (d2*)(A.xo + e));
  __builtin_nontemporal_store(x1 + a, (d2*)(A.xo + A.ld + e));
  __builtin_nontemporal_store(x0, (d2*)(A.obs + e));
  __builtin_nontemporal_store(x1, (d2*)(A.obs + A.ld + e));
  __builtin_nontemporal_store(a, (d2*)(A.obs + 2 * A.ld + e));
  __builtin_nontemporal_store(x0 * x1, (d2*)(A.rew + e));
  *(uint16_t*)(A.done + e) = (uint16_t)((x0.x > x1.x) | ((x0.y > x1.y) << 8));
}

int main(int argc, char** argv) {
  // launches per timed loop; the GPU needs ~50 ms of work to reach steady clocks: pass 3000 for the warm floor
  const int n_timed = argc > 1 ? atoi(argv[1]) : 300;
  const int64_t B = 1 << 20;
  const int NA = 64;
  double *x, *a, *xo, *obs, *rew;
  uint8_t* done;
  CK(hipMalloc(&x, 2 * B * 8));
  CK(hipMalloc(&a, (size_t)NA * B * 8));
  CK(hipMalloc(&xo, 2 * B * 8));
  CK(hipMalloc(&obs, 3 * B * 8));
  CK(hipMalloc(&rew, B * 8));
  CK(hipMalloc(&done, B));
  CK(hipMemset(x, 0, 2 * B * 8));
  CK(hipMemset(a, 0, (size_t)NA * B * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 73.0 * B;
  auto run = [&](const char* name, auto launch) -> int {
    for (int i = 0; i < n_timed / 10 + 20; ++i) launch(i);
    CK(hipDeviceSynchronize());
    const int n = n_timed;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.2f us/launch  %6.0f GB/s\n", name, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e9);
    return 0;
  };
  auto args = [&](int i) {
    // in-place state like the product (xo == x) -> x is re-read from cache/MALL next launch
    return Args{x, a + (size_t)(i % NA) * B, x, obs, rew, done, B};
  };
  run("K1 1 env/lane 8B one-shot", [&](int i) { hipLaunchKernelGGL(k1, dim3(B / 256), dim3(256), 0, 0, args(i)); });
  run("K2 2 env/lane 16B one-shot", [&](int i) { hipLaunchKernelGGL((k2<true, false>), dim3(B / 512), dim3(256), 0, 0, args(i)); });
  run("K2 no done row", [&](int i) { hipLaunchKernelGGL((k2<false, false>), dim3(B / 512), dim3(256), 0, 0, args(i)); });
  run("K2 nontemporal stores", [&](int i) { hipLaunchKernelGGL((k2<true, true>), dim3(B / 512), dim3(256), 0, 0, args(i)); });
  run("K4 4 env/lane one-shot", [&](int i) { hipLaunchKernelGGL(k4, dim3(B / 1024), dim3(256), 0, 0, args(i)); });
  for (int g : {256, 512, 1024, 2048}) {
    char nm[64];
    snprintf(nm, sizeof nm, "K5 persistent grid=%d", g);
    run(nm, [&](int i) { hipLaunchKernelGGL((k5<true>), dim3(g), dim3(256), 0, 0, args(i)); });
  }
  run("K5 persistent grid=1024 no done", [&](int i) { hipLaunchKernelGGL((k5<false>), dim3(1024), dim3(256), 0, 0, args(i)); });
  run("K6 1 env/lane, done as ballot bitmask", [&](int i) { hipLaunchKernelGGL(k6, dim3(B / 256), dim3(256), 0, 0, args(i)); });
  {  // padded layouts: one big arena, rows placed `ld` elements apart, arrays staggered by `stag` bytes
    double* arena;
    const int64_t maxld = B + 65536;
    CK(hipMalloc(&arena, (size_t)(8 * maxld + 8 * 65536) * 8 + (size_t)NA * B * 8));
    CK(hipMemset(arena, 0, (size_t)(8 * maxld + 8 * 65536) * 8));
    for (int64_t pad : {(int64_t)0, (int64_t)32, (int64_t)512, (int64_t)2080, (int64_t)33056}) {
      const int64_t ld = B + pad;
      double* px = arena;
      double* pxo = px;                 // in place like the product
      double* pobs = px + 2 * ld + pad; // staggered start
      double* prew = pobs + 3 * ld + pad;
      char nm[64];
      snprintf(nm, sizeof nm, "K2P nt, component stride B+%lld", (long long)pad);
      run(nm, [&](int i) {
        hipLaunchKernelGGL(k2p, dim3(B / 512), dim3(256), 0, 0,
                           ArgsP{px, a + (size_t)(i % NA) * B, pxo, pobs, prew, done, B, ld});
      });
    }
  }
  return 0;
}
