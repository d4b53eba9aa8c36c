// tools/gbs_lanes_bench.hip -- measurement only, not part of the product (round 4).
//
// The cstr's ignition front on the lanes that idle beside it (DESIGN section 0 row 2, section 8 item 5;
// tools/prototypes/gbs_lanes_cstr.py): explicit extrapolation, EIGHT LANES PER ENV, lane j integrates the big step H with
// Gragg's modified midpoint rule in n_j = 2 (j + 1) sub-steps (n_j + 1 evaluations of the product's own right-hand side,
// pcg_models.hpp), Aitken-Neville in h^2 over the eight lanes by cross-lane reads (order 16), error estimate
// T_87 - T_88 on the deepest lane.  One wave = 8 envs, one wave per workgroup (a SIMD to itself).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o _ab/gbs_lanes_bench.so tools/gbs_lanes_bench.hip
//   python tools/gbs_lanes_probe.py
#include "../pc-gym_amd/csrc/pcg_models.hpp"

#include <cstdio>

using M = pcg::Model<PCG_MODEL_CSTR>;
constexpr int NX = M::NX, LPE = 8;

struct Args {
  const double* x0;  // [NX][n]
  const double* u;   // [1][n]  (Tc)
  double* y;
  int* steps;        // [2][n]
  long long* clk;
  int n;
  double dt, tol, Ti, Caf, h0frac, facmax, safety;
  M::KP kp;
};

__global__ __launch_bounds__(64) void gbs_lanes(const Args A) {
  const int lane = threadIdx.x, g = lane >> 3, j = lane & 7, top = lane | 7;
  const int nj = 2 * (j + 1);
  int env = blockIdx.x * 8 + g;
  const bool real = env < A.n;
  if (!real) env = A.n - 1;
  const long long c0 = wall_clock64();
  double x[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = A.x0[(size_t)i * A.n + env];
  const double uu[3] = {A.u[env], A.Ti, A.Caf};
  const auto hold = M::hold<double>(A.kp, uu);
  double t = 0.0, H = A.dt * A.h0frac;
  int nacc = 0, nrej = 0;
  bool live = true;
  while (__any(live)) {
    if (live) {
      const double Hc = __builtin_fmin(H, A.dt - t);
      const double h = Hc / (double)nj, h2 = 2.0 * h;
      double z0[NX], z1[NX], f[NX];
      M::rhs(A.kp, hold, x, f);  // (the same on all eight lanes: the first evaluation of the big step)
#pragma unroll
      for (int i = 0; i < NX; ++i) z0[i] = x[i], z1[i] = __builtin_fma(h, f[i], x[i]);
      for (int s = 1; s < 2 * LPE; ++s) {  // the deepest lane (n = 16) sets the trip count of the wave
        if (s < nj) {
          M::rhs(A.kp, hold, z1, f);
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            const double zn = __builtin_fma(h2, f[i], z0[i]);
            z0[i] = z1[i], z1[i] = zn;
          }
        }
      }
      M::rhs(A.kp, hold, z1, f);
      double y[NX], prev[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] = 0.5 * (z0[i] + __builtin_fma(h, f[i], z1[i]));
      // Aitken-Neville in h^2: T_j <- T_j + (T_j - T_{j-1}) / ((n_j / n_{j-c})^2 - 1)
#pragma unroll
      for (int c = 1; c < LPE; ++c) {
        const double r = (double)(j + 1) / (double)(j + 1 - c > 0 ? j + 1 - c : 1);
        const double w = 1.0 / (r * r - 1.0 + (j >= c ? 0.0 : 1.0));
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const double up = __shfl_up(y[i], 1, LPE);
          if (c == LPE - 1) prev[i] = y[i];
          if (j >= c) y[i] = __builtin_fma(y[i] - up, w, y[i]);
        }
      }
      double en = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const double sc = A.tol + A.tol * __builtin_fmax(__builtin_fabs(x[i]), __builtin_fabs(y[i]));
        const double q = (y[i] - prev[i]) / sc;
        en = __builtin_fma(q, q, en);
      }
      en = __builtin_sqrt(en * (1.0 / NX));
      en = __shfl(en, top);
      if (!(en == en)) en = 1e10;
      const bool acc = en <= 1.0;
      double fac = A.safety * pow(__builtin_fmax(en, 1e-12), -1.0 / (2 * LPE - 1));
      fac = __builtin_fmin(acc ? A.facmax : 1.0, __builtin_fmax(0.1, fac));
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const double yt = __shfl(y[i], top);
        if (acc) x[i] = yt;
      }
      if (acc) {
        t += Hc;
        ++nacc;
      } else ++nrej;
      H = Hc * fac;
      live = (A.dt - t) > 1e-14 * A.dt && (nacc + nrej) < 100000;
    }
  }
  if (real && j == 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) A.y[(size_t)i * A.n + env] = x[i];
    A.steps[env] = nacc;
    A.steps[(size_t)A.n + env] = nrej;
  }
  const long long c1 = wall_clock64();
  if (lane == 0) {
    A.clk[blockIdx.x] = c0;
    A.clk[gridDim.x + blockIdx.x] = c1;
  }
}

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return (int)_e;                                                           \
    }                                                                           \
  } while (0)

// x0 [2][n], u [1][n], raw = the cstr's ten parameters; y [2][n], steps [2][n], wave_us [ceil(n/8)]
extern "C" __attribute__((visibility("default"))) int gbs_run(const double* x0, const double* u, int n, const double* raw, double dt, double tol,
                                                              double h0frac, double facmax, double safety, int reps, double* y, int* steps,
                                                              double* wave_us, double* kernel_us) {
  Args A{};
  double ddef[2];
  M::prep(raw, 0, 0, reinterpret_cast<double*>(&A.kp), ddef);
  A.n = n, A.dt = dt, A.tol = tol, A.Ti = ddef[0], A.Caf = ddef[1], A.h0frac = h0frac, A.facmax = facmax, A.safety = safety;
  const int waves = (n + 7) / 8;
  double *dx, *du, *dy;
  int* ds;
  long long* dc;
  CK(hipMalloc(&dx, sizeof(double) * NX * n));
  CK(hipMalloc(&du, sizeof(double) * n));
  CK(hipMalloc(&dy, sizeof(double) * NX * n));
  CK(hipMalloc(&ds, sizeof(int) * 2 * n));
  CK(hipMalloc(&dc, sizeof(long long) * 2 * waves));
  CK(hipMemcpy(dx, x0, sizeof(double) * NX * n, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, u, sizeof(double) * n, hipMemcpyHostToDevice));
  A.x0 = dx, A.u = du, A.y = dy, A.steps = ds, A.clk = dc;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(gbs_lanes, dim3(waves), dim3(64), 0, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gbs_lanes, dim3(waves), dim3(64), 0, 0, A);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  *kernel_us = 1e3 * ms / reps;
  CK(hipMemcpy(y, dy, sizeof(double) * NX * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(steps, ds, sizeof(int) * 2 * n, hipMemcpyDeviceToHost));
  long long* hc = new long long[2 * waves];
  CK(hipMemcpy(hc, dc, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost));
  for (int w = 0; w < waves; ++w) wave_us[w] = (hc[waves + w] - hc[w]) * 0.01;  // 100 MHz
  delete[] hc;
  (void)hipFree(dx), (void)hipFree(du), (void)hipFree(dy), (void)hipFree(ds), (void)hipFree(dc);
  return 0;
}
