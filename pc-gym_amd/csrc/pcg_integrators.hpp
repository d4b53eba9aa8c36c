// pcg_integrators.hpp -- per-lane one-step ODE integrators over one env step [0,dt]
// with the input held constant (zero-order hold; reference integrator.py:163-182
// builds integrator("cvodes", {x, p=u, ode}, 0, dt)).
//
//   rk4      classical RK4, n equal sub-steps.  State + 3 work vectors in VGPRs.
//   dopri5   Dormand-Prince 5(4) FSAL, per-lane adaptive step; semantic twin of the
//            reference's jax path (integrator.py:56-61: adaptive explicit 5(4) pair,
//            rtol=atol=1e-8, dt0=None).  Stage vectors k1..k7 live either in VGPRs
//            (RegStages) or in LDS laid out [stage][component][lane] (LdsStages) --
//            the variant BASELINE.json's north_star asks for the wide models.
//
// The controller is specified in DESIGN.md ("Adaptive stepping") and implemented
// independently in oracle/pcg_oracle.c.
#pragma once
#ifndef __HIPCC_RTC__  // built in under hipRTC
#include <hip/hip_runtime.h>
#endif

#include "pcg_models.hpp"

namespace pcg {

template <int NX, class F, class R>
PCG_DEV void rk4(const F& f, R (&x)[NX], double h, int nsub) {
  R k[NX], acc[NX], y[NX];
  const double h2 = 0.5 * h, h6 = h / 6.0;
  for (int s = 0; s < nsub; ++s) {
    f(x, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      acc[i] = k[i];
      y[i] = x[i] + h2 * k[i];
    }
    f(y, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      acc[i] = acc[i] + 2.0 * k[i];
      y[i] = x[i] + h2 * k[i];
    }
    f(y, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      acc[i] = acc[i] + 2.0 * k[i];
      y[i] = x[i] + h * k[i];
    }
    f(y, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x[i] + h6 * (acc[i] + k[i]);
  }
}

// ---- stage storage policies -------------------------------------------------
template <int NX>
struct RegStages {
  double k[6][NX];
  PCG_DEV double get(int s, int i) const { return k[s][i]; }
  PCG_DEV void set(int s, int i, double v) { k[s][i] = v; }
};

// LDS layout [stage][component][thread]: lane-contiguous 8-byte words, so each
// ds_read_b64 / ds_write_b64 of a wave touches 512 contiguous bytes (conflict-free).
template <int NX, int THREADS>
struct LdsStages {
  double* base;  // &lds[threadIdx.x]
  PCG_DEV double get(int s, int i) const { return base[(s * NX + i) * THREADS]; }
  PCG_DEV void set(int s, int i, double v) { base[(s * NX + i) * THREADS] = v; }
  static constexpr size_t bytes() { return sizeof(double) * 6 * NX * THREADS; }
};

// 1/x for the error norm: hardware reciprocal estimate + one Newton step (callers guarantee a finite, positive,
// normal x).  The norm only gates accept/reject and steers h; ~1e-9 relative is ample.
PCG_DEV double fast_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// Step-size factors are QUANTISED: only the sign, the exponent and the top 6 mantissa bits are kept (truncation, a
// grid of 0.8-1.6 % relative spacing).  A step-size controller is indifferent to a 1 % change of its factor, and the
// grid makes the factor -- hence the whole sequence of step sizes -- independent of how E^(-1/5) was evaluated: the
// kernel's fp32 log2/exp2 units (~5e-7 relative), the oracle's double pow() and any later implementation produce
// bit-identical step sequences, which is what lets the parity tests compare the adaptive path at 1e-12.
PCG_DEV double qtrunc6(double v) {
  return __longlong_as_double(__double_as_longlong(v) & ~((1LL << 46) - 1));
}
// qtrunc6(scale * E2^(-1/10)): E2 = mean square of the scaled error (E = sqrt(E2), so this is scale * E^(-1/5)).
// Fast path through the fp32 log2/exp2 units (2 transcendental + 3 conversion/multiply instructions instead of ~60
// for a double log + exp); when the fp32 result lands within 2^-10 of a grid cell edge (0.2 % of the calls), or E2 is
// outside the comfortable fp32 range, the double path decides, so the quantised value never depends on the fp32
// rounding.  Both infinities map to values the callers clip (E2 -> 0: huge, E2 -> inf: tiny).
// NEG_EXP: the exponent applied to the MEAN SQUARE, i.e. half the method's: 0.1 for the 5(4) pair (E^(-1/5)),
// 1/6 for the 3(2) Rosenbrock pair (E^(-1/3)).
PCG_DEV double ctrl_pow_e(double E2, double scale, float neg_exp_f, double neg_exp_d) {
  double f = scale * (double)__builtin_amdgcn_exp2f(-neg_exp_f * __builtin_amdgcn_logf((float)E2));
  const unsigned frac = (unsigned)(__double_as_longlong(f) >> 36) & 0x3FFu;  // the 10 bits below the kept ones
  if (frac == 0u || frac == 0x3FFu || !(E2 > 1e-30 && E2 < 1e30))
    f = scale * exp_bounded(-neg_exp_d * log_pos(E2));
  return qtrunc6(f);
}
PCG_DEV double ctrl_pow(double E2, double scale) { return ctrl_pow_e(E2, scale, 0.1f, 0.1); }

// Linear combinations of stage derivatives as EXPLICIT fused multiply-adds in a fixed order.  The functions that use
// them switch compiler contraction off (#pragma clang fp contract(off)), so the arithmetic that feeds back into the
// state -- stage points, the new solution, the step size -- is one exactly specified sequence of IEEE operations:
// every kernel that integrates an env (classic, work-queue, fused rollout) produces the same bits, and the oracle
// (C fma()) can follow it bit for bit.  That matters for the stability-limited extraction model, whose step-size
// sequence amplifies a last-bit difference into a different (equally valid) sequence (tests/helpers.py).
PCG_DEV double lc1(double c1, double k1) { return c1 * k1; }
PCG_DEV double lc2(double c1, double k1, double c2, double k2) { return __builtin_fma(c2, k2, c1 * k1); }
PCG_DEV double lc3(double c1, double k1, double c2, double k2, double c3, double k3) {
  return __builtin_fma(c3, k3, lc2(c1, k1, c2, k2));
}
PCG_DEV double lc4(double c1, double k1, double c2, double k2, double c3, double k3, double c4, double k4) {
  return __builtin_fma(c4, k4, lc3(c1, k1, c2, k2, c3, k3));
}
PCG_DEV double lc5(double c1, double k1, double c2, double k2, double c3, double k3, double c4, double k4, double c5,
                   double k5) {
  return __builtin_fma(c5, k5, lc4(c1, k1, c2, k2, c3, k3, c4, k4));
}
PCG_DEV double lc6(double c1, double k1, double c2, double k2, double c3, double k3, double c4, double k4, double c5,
                   double k5, double c6, double k6) {
  return __builtin_fma(c6, k6, lc5(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5));
}
// x + h * s
PCG_DEV double axpy(double h, double s, double x) { return __builtin_fma(h, s, x); }

// mean square of v_i / (atol + rtol max(|y0_i|, |y1_i|))  (the RMS norm squared)
template <int NX>
PCG_DEV double ms_scaled(const double (&v)[NX], const double (&y0)[NX], const double (&y1)[NX], int n,
                         double rtol, double atol) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const double sc = atol + rtol * fmax(fabs(y0[i]), fabs(y1[i]));
    const double r = v[i] * fast_rcp(sc);  // sc > 0
    s += (i < n) ? r * r : 0.0;
  }
  return s / n;
}
template <int NX>
PCG_DEV double rms_scaled(const double (&v)[NX], const double (&y0)[NX], const double (&y1)[NX], int n,
                          double rtol, double atol) {
  return sqrt(ms_scaled<NX>(v, y0, y1, n, rtol, atol));
}

// returns 0 ok, 1 step budget exhausted, 2 step-size underflow
template <int NX, class F, class ST>
PCG_DEV int dopri5(const F& f, ST& K, double (&x)[NX], int n, double dt, double rtol, double atol,
                   int max_steps, int& nacc, int& nrej) {
#pragma clang fp contract(off)
  constexpr double a21 = 1.0 / 5;
  constexpr double a31 = 3.0 / 40, a32 = 9.0 / 40;
  constexpr double a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9;
  constexpr double a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729;
  constexpr double a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176,
                   a65 = -5103.0 / 18656;
  constexpr double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
  constexpr double e1 = 71.0 / 57600, e3 = -71.0 / 16695, e4 = 71.0 / 1920, e5 = -17253.0 / 339200,
                   e6 = 22.0 / 525, e7 = -1.0 / 40;
  double y[NX], kk[NX], w[NX];
  int acc = 0, rej = 0, status = 0;

  f(x, kk);
#pragma unroll
  for (int i = 0; i < NX; ++i) K.set(0, i, kk[i]);
  // initial step size (Hairer, Norsett & Wanner, II.4)
  double h;
  {
    const double d0 = rms_scaled<NX>(x, x, x, n, rtol, atol);
    const double d1 = rms_scaled<NX>(kk, x, x, n, rtol, atol);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h0 = fmin(h0, dt);
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] = axpy(h0, kk[i], x[i]);
    f(y, w);
#pragma unroll
    for (int i = 0; i < NX; ++i) w[i] -= kk[i];
    const double d2 = rms_scaled<NX>(w, x, x, n, rtol, atol) / h0;
    const double dm = fmax(d1, d2);
    const double h1 = (dm <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : ctrl_pow(dm * dm * 1e4, 1.0);
    h = fmin(qtrunc6(fmin(100.0 * h0, h1)), dt);
  }
  double t = 0.0;
  bool rejected_last = false;
  for (;;) {
    bool last = false;
    if (acc + rej >= max_steps) {
      status = 1;
      break;
    }
    if (t + h >= dt * (1.0 - 1e-14)) {
      h = dt - t;
      last = true;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] = axpy(h, lc1(a21, K.get(0, i)), x[i]);
    f(y, kk);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      K.set(1, i, kk[i]);
      y[i] = axpy(h, lc2(a31, K.get(0, i), a32, kk[i]), x[i]);
    }
    f(y, kk);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      K.set(2, i, kk[i]);
      y[i] = axpy(h, lc3(a41, K.get(0, i), a42, K.get(1, i), a43, kk[i]), x[i]);
    }
    f(y, kk);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      K.set(3, i, kk[i]);
      y[i] = axpy(h, lc4(a51, K.get(0, i), a52, K.get(1, i), a53, K.get(2, i), a54, kk[i]), x[i]);
    }
    f(y, kk);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      K.set(4, i, kk[i]);
      y[i] = axpy(h, lc5(a61, K.get(0, i), a62, K.get(1, i), a63, K.get(2, i), a64, K.get(3, i), a65, kk[i]), x[i]);
    }
    f(y, kk);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      K.set(5, i, kk[i]);
      y[i] = axpy(h, lc5(b1, K.get(0, i), b3, K.get(2, i), b4, K.get(3, i), b5, K.get(4, i), b6, kk[i]), x[i]);
    }
    f(y, kk);  // k7 at the 5th-order solution (FSAL)
#pragma unroll
    for (int i = 0; i < NX; ++i)
      w[i] = h * lc6(e1, K.get(0, i), e3, K.get(2, i), e4, K.get(3, i), e5, K.get(4, i), e6, K.get(5, i), e7, kk[i]);
    const double E2 = ms_scaled<NX>(w, x, y, n, rtol, atol);  // accept iff E = sqrt(E2) < 1
    // one evaluation of the controller for both outcomes, the outcome applied by selects (in a wave of 64 lanes some
    // lane rejects in almost every iteration, so an if / else with the controller in both arms executed both; the
    // values are the same, bit for bit)
    const bool ok = E2 < 1.0;
    double fac = (E2 == E2) ? fmax(0.2, ctrl_pow(E2, 0.9)) : 0.2;  // NaN -> hardest shrink
    fac = fmin(ok ? (rejected_last ? 1.0 : 10.0) : 1.0, fac);
    t = ok ? t + h : t;
    h *= fac;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      x[i] = ok ? y[i] : x[i];
      K.set(0, i, ok ? kk[i] : K.get(0, i));
    }
    rejected_last = !ok;
    acc += ok ? 1 : 0;
    rej += ok ? 0 : 1;
    if (ok && last) break;
    if (!ok && !(h > 1e-13 * dt)) {  // step-size underflow (NaN state / blow-up): give up on this lane
      status = 2;
      break;
    }
  }
  nacc = acc;
  nrej = rej;
  return status;
}

// ---------------------------------------------------------------------------------------------------------------
// Rodas3 -- a stiff-capable integrator (the reference solves with CVODES BDF,
// integrator.py:163-182).  Sandu, Verwer, Blom, Spee, Carmichael & Potra (1997): 4-stage linearly implicit
// Rosenbrock method, order 3 with an embedded order-2 estimate, L-stable, stiffly accurate, gamma = 1/2; per step one
// Jacobian (forward differences of the model's RHS: generic over the registry), one LU factorisation of
// W = I/(gamma h) - J with partial pivoting and four triangular solves.  Per-lane adaptive step size with the same
// norm, quantised factor and failure semantics as dopri5() (exponent 1/3, growth limit 6).
// The per-lane matrix lives in LDS  W[(i*NX + j)][lane]  (lane-contiguous 8-byte words: conflict-free), the pivot
// rows behind it; vectors stay in registers.  For the registry models at their canonical stiffness the explicit pair
// is faster; this integrator is for the user whose model is stiff beyond |lambda| dt ~ 1e3.
// ---------------------------------------------------------------------------------------------------------------
// one wave per workgroup; past 16 states only half of its lanes carry an env, so that the matrices (nx^2 x lanes x 8 B)
// still fit the 160 KB of LDS (nx = 24: 147 KB + pivots)
constexpr int ros_threads(int nx) { return nx <= 16 ? 64 : 32; }
constexpr size_t ros_lds_doubles(int nx) { return (size_t)(nx * nx + (nx + 1) / 2) * ros_threads(nx); }

template <int NX>
struct RosLds {
  static constexpr int T = ros_threads(NX);
  double* W;     // &lds[lane]
  int32_t* piv;  // &((int32_t*)(lds + NX*NX*T))[lane]
  PCG_DEV explicit RosLds(double* lds) : W(lds + threadIdx.x), piv(reinterpret_cast<int32_t*>(lds + NX * NX * T) + threadIdx.x) {}
  PCG_DEV double& w(int i, int j) const { return W[(i * NX + j) * T]; }
  PCG_DEV int32_t& p(int k) const { return piv[k * T]; }
};

// unroll depth of the factorisation's outer loops: full for the small systems (the compiler can then batch the LDS
// reads of a row instead of paying one LDS round trip per element), rolled for the large ones (code size)
constexpr int ros_unroll(int nx) { return nx <= 10 ? nx : 1; }

// LU with partial pivoting, in place; returns false on a (numerically) singular pivot
template <int NX>
PCG_DEV bool ros_lu(const RosLds<NX>& L, int n) {
  bool ok = true;
  constexpr int UK = ros_unroll(NX);
#pragma unroll UK
  for (int k = 0; k < NX; ++k) {
    if (k >= n) break;
    int pk = k;
    double best = fabs(L.w(k, k));
#pragma unroll
    for (int i = k + 1; i < NX; ++i) {
      if (i < n) {
        const double v = fabs(L.w(i, k));
        pk = (v > best) ? i : pk;
        best = (v > best) ? v : best;
      }
    }
    L.p(k) = pk;
    if (pk != k) {
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        if (j < n) {
          const double a = L.w(k, j), b = L.w(pk, j);
          L.w(k, j) = b;
          L.w(pk, j) = a;
        }
      }
    }
    const double d = L.w(k, k);
    ok = ok && (fabs(d) > 1e-300) && (d == d);
    const double inv = 1.0 / d;
    double rowk[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) rowk[j] = (j > k && j < n) ? L.w(k, j) : 0.0;
#pragma unroll UK
    for (int i = k + 1; i < NX; ++i) {
      if (i >= n) break;
      const double l = L.w(i, k) * inv;
      L.w(i, k) = l;
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        if (j > k && j < n) L.w(i, j) = L.w(i, j) - l * rowk[j];
      }
    }
  }
  return ok;
}

// solve W z = b in place (b in registers; the row swaps by select chains: no dynamic register indexing)
template <int NX>
PCG_DEV void ros_solve(const RosLds<NX>& L, int n, double (&b)[NX]) {
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    if (k < n) {
      const int pk = L.p(k);
      const double bk = b[k];
      double bp = bk;
#pragma unroll
      for (int i = 0; i < NX; ++i) bp = (i == pk) ? b[i] : bp;
#pragma unroll
      for (int i = 0; i < NX; ++i) b[i] = (i == pk) ? bk : b[i];
      b[k] = bp;
    }
  }
#pragma unroll
  for (int i = 1; i < NX; ++i) {
    if (i < n) {
      double sacc = b[i];
#pragma unroll
      for (int j = 0; j < NX; ++j)
        if (j < i) sacc -= L.w(i, j) * b[j];
      b[i] = sacc;
    }
  }
#pragma unroll
  for (int ii = 0; ii < NX; ++ii) {
    const int i = NX - 1 - ii;
    if (i < n) {
      double sacc = b[i];
#pragma unroll
      for (int j = 0; j < NX; ++j)
        if (j > i && j < n) sacc -= L.w(i, j) * b[j];
      b[i] = sacc / L.w(i, i);
    }
  }
}

// returns PCG_ST_OK, PCG_ST_MAX_STEPS or PCG_ST_UNDERFLOW (the caller poisons the state on failure)
template <int NX, class F>
PCG_DEV int rodas3(const F& f, const RosLds<NX>& L, double (&x)[NX], int n, double dt, double rtol, double atol,
                   int max_steps, int& nacc, int& nrej) {
#pragma clang fp contract(off)
  constexpr double gam = 0.5;
  double f0[NX], k1[NX], k2[NX], k3[NX], k4[NX], y[NX], fy[NX];
  int acc = 0, rej = 0, status = 0;
  f(x, f0);
  double h;
  {  // initial step: h0 of Hairer, Norsett & Wanner II.4 (the first stage of dopri5()'s heuristic), quantised
    const double d0 = rms_scaled<NX>(x, x, x, n, rtol, atol);
    const double d1 = rms_scaled<NX>(f0, x, x, n, rtol, atol);
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    h = fmin(qtrunc6(100.0 * h0), dt);
  }
  double t = 0.0;
  bool rejected_last = false;
  for (;;) {
    bool last = false;
    if (acc + rej >= max_steps) {
      status = 1;
      break;
    }
    if (t + h >= dt * (1.0 - 1e-14)) {
      h = dt - t;
      last = true;
    }
    // Jacobian by forward differences, W = I/(gamma h) - J
    // perturbation ~ sqrt(eps) / rtol x the error weight of the component (CVODES' difference quotient scales the
    // same way); the x_max term only keeps it non-zero when atol = 0 and x_j = 0
    double xmax = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) xmax = fmax(xmax, (i < n) ? fabs(x[i]) : 0.0);
    const double wfloor = atol / rtol + 1e-12 * xmax + 1e-100;
#pragma unroll 1
    for (int j = 0; j < NX; ++j) {
      if (j >= n) break;
      double xj = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) xj = (i == j) ? x[i] : xj;
      const double del = 1.4901161193847656e-8 * (fabs(xj) + wfloor);
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] = (i == j) ? x[i] + del : x[i];
      f(y, fy);
      const double idel = 1.0 / ((xj + del) - xj);  // the perturbation that was actually applied
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (i < n) L.w(i, j) = -(fy[i] - f0[i]) * idel;
    }
    const double igh = 1.0 / (gam * h), ih = 1.0 / h;
    for (int i = 0; i < NX; ++i) {
      if (i >= n) break;
      L.w(i, i) = L.w(i, i) + igh;
    }
    const bool lu_ok = ros_lu<NX>(L, n);
    // stage 1 .. 4 (a21 = 0: stage 2 re-uses f(x))
#pragma unroll
    for (int i = 0; i < NX; ++i) k1[i] = f0[i];
    ros_solve<NX>(L, n, k1);
#pragma unroll
    for (int i = 0; i < NX; ++i) k2[i] = f0[i] + (4.0 * ih) * k1[i];
    ros_solve<NX>(L, n, k2);
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] = x[i] + 2.0 * k1[i];
    f(y, fy);
#pragma unroll
    for (int i = 0; i < NX; ++i) k3[i] = fy[i] + ih * (k1[i] - k2[i]);
    ros_solve<NX>(L, n, k3);
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] = x[i] + 2.0 * k1[i] + k3[i];
    f(y, fy);
#pragma unroll
    for (int i = 0; i < NX; ++i) k4[i] = fy[i] + ih * (k1[i] - k2[i] - (8.0 / 3.0) * k3[i]);
    ros_solve<NX>(L, n, k4);
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] = x[i] + 2.0 * k1[i] + k3[i] + k4[i];  // m = (2, 0, 1, 1); error = k4
    double E2 = ms_scaled<NX>(k4, x, y, n, rtol, atol);
    if (!lu_ok) E2 = __builtin_nan("");
    if (E2 < 1.0) {
      double fac = fmin(6.0, fmax(0.2, ctrl_pow_e(E2, 0.9, 1.0f / 6.0f, 1.0 / 6.0)));
      if (rejected_last && fac > 1.0) fac = 1.0;
      t += h;
      h *= fac;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = y[i];
      rejected_last = false;
      ++acc;
      if (last) break;
      f(x, f0);
    } else {
      double fac = (E2 == E2) ? fmax(0.2, ctrl_pow_e(E2, 0.9, 1.0f / 6.0f, 1.0 / 6.0)) : 0.2;
      if (fac > 1.0) fac = 1.0;
      h *= fac;
      rejected_last = true;
      ++rej;
      if (!(h > 1e-13 * dt)) {
        status = 2;
        break;
      }
    }
  }
  nacc = acc;
  nrej = rej;
  return status;
}

}  // namespace pcg
