// tools/seulex_lanes_bench.hip -- measurement only, not part of the product (round 4).
//
// What does the heaviest env of a Rodas4 launch cost when its chain is spread over the lanes that idle beside it?
// (tools/prototypes/seulex_lanes_me10.py.)  Extrapolated linearly implicit Euler with a fixed
// column on the 10-state extraction cascade: EIGHT LANES PER ENV, lane j integrates the big step H with n_j = j + 1
// sub-steps (I / h_j - J) d = f(y), J frozen at the start of the big step -- each lane factors its own structured
// W = theta_j I - J with the model's ros_factor / ros_solve (pcg_models.hpp, the product's own device functions) -- then the
// Aitken-Neville tableau over the eight lanes by cross-lane reads; error estimate T_87 - T_88 on the deepest lane,
// elementary step-size controller.  One wave = 8 envs; one wave per workgroup so that every wave has a SIMD to itself,
// as the heavy wave of a work-queue tile nearly has (s_setprio).  Every wave stamps the 100 MHz clock at start and end.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o _ab/seulex_lanes_bench.so tools/seulex_lanes_bench.hip
//   python tools/seulex_lanes_probe.py
#include "../pc-gym_amd/csrc/pcg_models.hpp"

#include <cstdio>

using M = pcg::Model<pcg::PCG_KID_ME_SQ>;
constexpr int NX = M::NX, LPE = 8;  // lanes per env

struct Args {
  const double* x0;  // [NX][n]
  const double* u;   // [2][n]  (L, G)
  double* y;         // [NX][n]
  int* steps;        // [2][n] accepted, rejected big steps
  long long* clk;    // [2][waves]
  int n;
  double dt, tol, X0, Y6, h0frac, facmax, safety;
  M::KP kp;
};

__global__ __launch_bounds__(64) void seulex_lanes(const Args A) {
  const int lane = threadIdx.x, g = lane >> 3, j = lane & 7, top = lane | 7;
  const int nj = j + 1;
  int env = blockIdx.x * 8 + g;
  const bool real = env < A.n;
  if (!real) env = A.n - 1;
  const long long c0 = wall_clock64();
  double x[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = A.x0[(size_t)i * A.n + env];
  const double uu[4] = {A.u[env], A.u[(size_t)A.n + env], A.X0, A.Y6};
  const auto hold = M::hold<double>(A.kp, uu);
  double t = 0.0, H = A.dt * A.h0frac;
  int nacc = 0, nrej = 0;
  bool live = true;
  while (__any(live)) {
    if (live) {
      const double Hc = __builtin_fmin(H, A.dt - t);
      const double theta = (double)nj / Hc;
      M::RosFac F;
      M::ros_factor(A.kp, hold, x, theta, F);
      double y[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] = x[i];
      for (int s = 0; s < LPE; ++s) {  // the deepest lane sets the trip count of the wave; shallower lanes idle
        if (s < nj) {
          double f[NX];
          M::rhs(A.kp, hold, y, f);
          M::ros_solve(F, f);
#pragma unroll
          for (int i = 0; i < NX; ++i) y[i] += f[i];
        }
      }
      // Aitken-Neville in h: column c, T_j <- T_j + (T_j - T_{j-1}) (n_j - c) / c ... with n_j / n_{j-c} - 1 = c / (n_j - c)
      double prev[NX];
#pragma unroll
      for (int c = 1; c < LPE; ++c) {
        const double w = (double)(nj - c) / (double)c;
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
      const unsigned long long okm = __ballot(F.ok);  // every lane's factorisation must have had positive pivots
      const bool okg = ((okm >> (lane & ~7)) & 0xFFull) == 0xFFull;
      en = __shfl(en, top);
      if (!okg || !(en == en)) en = 1e10;
      const bool acc = en <= 1.0;
      double fac = A.safety * pow(__builtin_fmax(en, 1e-12), -1.0 / LPE);
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

// x0 [NX][n], u [2][n], raw = Vl,Vg,m,Kla,eq_exponent,X0,Y6 (eq_exponent must be 2); y [NX][n], steps [2][n],
// wave_us [ceil(n/8)] (duration of each wave), kernel_us = mean launch duration by events over `reps` launches
extern "C" __attribute__((visibility("default"))) int seulex_run(const double* x0, const double* u, int n, const double* raw, double dt,
                                                                 double tol, double h0frac, double facmax, double safety, int reps, double* y, int* steps, double* wave_us, double* kernel_us) {
  Args A{};
  double ddef[2];
  M::prep(raw, 0, 0, reinterpret_cast<double*>(&A.kp), ddef);
  A.n = n, A.dt = dt, A.tol = tol, A.X0 = ddef[0], A.Y6 = ddef[1], A.h0frac = h0frac, A.facmax = facmax, A.safety = safety;
  const int waves = (n + 7) / 8;
  double *dx, *du, *dy;
  int* ds;
  long long* dc;
  CK(hipMalloc(&dx, sizeof(double) * NX * n));
  CK(hipMalloc(&du, sizeof(double) * 2 * n));
  CK(hipMalloc(&dy, sizeof(double) * NX * n));
  CK(hipMalloc(&ds, sizeof(int) * 2 * n));
  CK(hipMalloc(&dc, sizeof(long long) * 2 * waves));
  CK(hipMemcpy(dx, x0, sizeof(double) * NX * n, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, u, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  A.x0 = dx, A.u = du, A.y = dy, A.steps = ds, A.clk = dc;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(seulex_lanes, dim3(waves), dim3(64), 0, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(seulex_lanes, dim3(waves), dim3(64), 0, 0, A);
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
