// pcg_models.hpp -- ODE right-hand sides of the hot-path models as device functors.
//
// One struct per model: compile-time sizes, a POD of pre-folded constants (KP, read
// through wave-uniform scalar loads -> SGPRs), a per-env-step `Hold` of everything
// that depends only on the held input u (zero-order hold over [0,dt],
// reference integrator.py:163-182), and rhs(kp, hold, x, dx).
//
// The arithmetic restates reference src/pcgym/model_classes.py (line ranges per
// model below); constants that the reference recomputes on every call
// (q/V, 1/(rho*C), ...) are folded once on the host in prep().
#pragma once
#ifndef __HIPCC_RTC__  // built in under hipRTC
#include <hip/hip_runtime.h>
#endif

#include "../../include/pcgym_hip.h"
#include "pcg_pack.hpp"

namespace pcg {

// model code is written once for R = double (one env per lane) and R = Pack<W> (W envs per lane);
// bring the scalar math functions into this namespace next to the Pack overloads
using ::exp;
using ::fabs;
using ::log;
using ::pow;
using ::sqrt;

#define PCG_DEV __device__ __forceinline__
#define PCG_HD __host__ __device__ inline
// Plan constants are read through the CONSTANT address space (AMDGPU addrspace 4): a load with a
// wave-uniform address then always becomes a scalar s_load (SGPR destination, scalar cache), even
// after the kernel has issued vector stores -- with a plain global pointer the compiler must assume
// the stores may alias and falls back to per-lane global_load (51 of them in the first build).
#define PCG_CONSTANT __attribute__((address_space(4)))

template <int ID>
struct Model;

// helpers of the models' end-point exponent hooks (PCG_INT_RODAS4, pcg_integrators.hpp)
PCG_DEV int ep_trunc(double v, int kmax) { return (v > 0.0) ? (int)__builtin_fmin(v, (double)kmax) : 0; }
// floor(log2(v)) for v >= 1 (0 below): exponent extraction, exact
PCG_DEV int ep_ilog2(double v) {
  return (v >= 1.0) ? (int)((__double_as_longlong(v) >> 52) & 0x7FF) - 1023 : 0;
}

// piecewise-linear log2 of a positive normal number: exponent + (mantissa - 1), mantissa in [1, 2).  Exponent extraction,
// one exact subtraction and one addition: the oracle's frexp() twin gives the same bits (the cooperative rule of
// PCG_INT_RODAS4 plans, M::coop_key, must pick the same envs on both sides)
PCG_DEV double plog2(double v) {
  const long long b = __double_as_longlong(v);
  const int e = (int)((b >> 52) & 0x7FF) - 1023;
  const double m = __longlong_as_double((b & 0x000FFFFFFFFFFFFFLL) | 0x3FF0000000000000LL);
  return (double)e + (m - 1.0);
}

// ---------------------------------------------------------------------------
// cstr -- model_classes.py:23-62.  raw = q,V,rho,C,deltaHr,EA_over_R,k0,UA,Ti,Caf
// u = [Tc | Ti, Caf]
// ---------------------------------------------------------------------------
template <>
struct Model<PCG_MODEL_CSTR> {
  static constexpr int NX = 2, NA = 1, NDM = 2, NRAW = 10;
  static constexpr bool DYNAMIC = false;
  static constexpr bool FULL = true;  // all kernel specialisations
  struct KP {
    double qV, c1, c2, k0, nEAR, iEA;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R Tc, Ti, Caf;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int /*nx*/, int /*nu*/, double* kp_out, double* ddef) {
    KP k;
    k.qV = r[0] / r[1];
    k.c1 = (-r[4]) * (1.0 / (r[2] * r[3]));
    k.c2 = r[7] * (1.0 / (r[2] * r[3] * r[1]));
    k.k0 = r[6];
    k.nEAR = -r[5];
    k.iEA = 1.0 / r[5];
    __builtin_memcpy(kp_out, &k, sizeof(k));
    ddef[0] = r[8];
    ddef[1] = r[9];
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0], u[1], u[2]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R ca = x[0], T = x[1];
    // Arrhenius factor: the exponent -EA/(R T) lies in (-700, 0) for any T > 12.5 K, so the bounded
    // exp and the Newton divide apply (quotient to ~0.5 ulp: the exponent is O(30), an ulp there is
    // 3e-15 on rA, which the T balance amplifies ~100x by cancellation -- parity bar 1e-12)
    const R rA = k.k0 * exp_bounded(div_fast(k.nEAR, T)) * ca;
    dx[0] = k.qV * (h.Caf - ca) - rA;
    dx[1] = k.qV * (h.Ti - T) + k.c1 * rA + k.c2 * (h.Tc - T);
  }
  // Guard of the fixed-step plans (PCG_INT_RK4G, PCG_INT_T5G): g = d(dT/dt)/dT = c1 rA EA/T^2 - (q/V + c2), the growth
  // rate of the thermal feedback (> 0: the reaction heats itself faster than flow and jacket cool it -- ignition, errors
  // amplify), and rho = k + c1 rA EA/T^2 + q/V + c2, a bound of the fastest rate.  EA/T^2 = z^2/EA with z = -EA/T, the
  // Arrhenius exponent the right-hand side computes anyway: no second division.  Calibrated on (state, input) pairs of
  // episodes over the whole observation box (tests/test_rk4g.py, tests/test_erk.py, tools/prototypes/cstr_guard_t5.py):
  // every env with g <= 0 and rho h below the scheme's limit at the checked states is within 7.5e-7 of a 1e-13 solve; the
  // canonical closed loop (T <= 330 K) never trips it, the ignition branch and the hot branch always do.
  static constexpr bool GUARD = true;
  template <class R, class K>
  PCG_DEV static void guard(const K& k, const HoldT<R>&, const R (&x)[NX], R& g, R& rho) {
    const R ca = x[0], T = x[1];
    const R z = div_fast(k.nEAR, T);
    const R kk = k.k0 * exp_bounded(z);
    const R fb = ((k.c1 * (kk * ca)) * (z * z)) * k.iEA;
    const double base = k.qV + k.c2;
    g = fb - base;
    rho = kk + fb + base;
  }
  // rhs() and guard() at the same point in one pass (one Arrhenius factor for both): the same bits as the two calls
  template <class R, class K>
  PCG_DEV static void rhs_guard(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX], R& g, R& rho) {
    const R ca = x[0], T = x[1];
    const R z = div_fast(k.nEAR, T);
    const R kk = k.k0 * exp_bounded(z);
    const R rA = kk * ca;
    dx[0] = k.qV * (h.Caf - ca) - rA;
    dx[1] = k.qV * (h.Ti - T) + k.c1 * rA + k.c2 * (h.Tc - T);
    const R fb = ((k.c1 * rA) * (z * z)) * k.iEA;
    const double base = k.qV + k.c2;
    g = fb - base;
    rho = kk + fb + base;
  }
};

// ---------------------------------------------------------------------------
// four_tank -- model_classes.py:864-931.  raw = g,gamma_1,gamma_2,k1,k2,a1..a4,A1..A4
// ---------------------------------------------------------------------------
template <>
struct Model<PCG_MODEL_FOUR_TANK> {
  static constexpr int NX = 4, NA = 2, NDM = 0, NRAW = 13;
  static constexpr bool DYNAMIC = false;
  static constexpr bool FULL = true;  // all kernel specialisations
  struct KP {
    double g2;              // 2 g
    double o1, o2, o3, o4;  // a_i / A_i
    double i31, i42;        // a3/A1, a4/A2
    double p1, p2, p3, p4;  // pump gains
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R q1, q2, q3, q4;  // pump inflow terms, constant over the step
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k;
    k.g2 = 2 * r[0];
    k.o1 = r[5] / r[9];
    k.o2 = r[6] / r[10];
    k.o3 = r[7] / r[11];
    k.o4 = r[8] / r[12];
    k.i31 = r[7] / r[9];
    k.i42 = r[8] / r[10];
    k.p1 = (r[1] * r[3]) / r[9];
    k.p2 = (r[2] * r[4]) / r[10];
    k.p3 = ((1 - r[2]) * r[4]) / r[11];
    k.p4 = ((1 - r[1]) * r[3]) / r[12];
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    return HoldT<R>{k.p1 * u[0], k.p2 * u[1], k.p3 * u[1], k.p4 * u[0]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    // tank levels are O(0.01..1) m: the range-restricted square root (pcg_pack.hpp) applies
    const R s1 = sqrt_pos(k.g2 * x[0]), s2 = sqrt_pos(k.g2 * x[1]);
    const R s3 = sqrt_pos(k.g2 * x[2]), s4 = sqrt_pos(k.g2 * x[3]);
    dx[0] = -k.o1 * s1 + k.i31 * s3 + h.q1;
    dx[1] = -k.o2 * s2 + k.i42 * s4 + h.q2;
    dx[2] = -k.o3 * s3 + h.q3;
    dx[3] = -k.o4 * s4 + h.q4;
  }
};

// Y^e/m for the extraction models: e == 2 (the reference default, model_classes.py:365) is the hot case
// and is a multiply (bit-identical to pow(Y, 2.0)); any other exponent goes through pow().  The choice is a
// COMPILE-TIME parameter: with a run-time branch the ~250-instruction pow sits between the five stages of
// every RHS evaluation, splits them into separate basic blocks and inflates the DOPRI5 loop to 70 KB of code.
// Plans with eq_exponent == 2 are routed to the *_SQ instantiations (internal ids below).
template <bool SQ, class R>
PCG_DEV R eq_curve(const R& Y, double e, double inv_m) {
  if constexpr (SQ) return (Y * Y) * inv_m;
  else return pow(Y, e) * inv_m;
}
// internal kernel-table ids behind the public enum pcg_model
constexpr int PCG_KID_ME_SQ = PCG_MODEL_COUNT, PCG_KID_ME_REACTIVE_SQ = PCG_MODEL_COUNT + 1,
              PCG_KID_COUNT = PCG_MODEL_COUNT + 2;

// ---------------------------------------------------------------------------
// multistage_extraction -- model_classes.py:346-430.  raw = Vl,Vg,m,Kla,eq_exponent,X0,Y6
// u = [L, G | X0, Y6];  x = X1,Y1,...,X5,Y5
// ---------------------------------------------------------------------------
// the cooperative phase of PCG_INT_RODAS4 plans (pcg_seulex.hpp) is calibrated for the eq_exponent == 2 cascade only
template <bool SQ>
struct MECoop {};
template <>
struct MECoop<true> {
  static constexpr bool COOP = true;
};
template <bool SQ>
struct MEImpl : MECoop<SQ> {
  static constexpr int NX = 10, NA = 2, NDM = 2, NRAW = 7;
  static constexpr bool DYNAMIC = false;
  static constexpr bool FULL = true;  // all kernel specialisations
  struct KP {
    double iVl, iVg, inv_m, KlaVl, e;
    double klap, eK;  // Kla Vl / Vl and Kla Vl / Vg, folded for rhs() and ros_factor()
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R L, G, X0, Y6;
    R a, c;  // through-flow rates L / Vl, G / Vg: held over the env step, so the divisions by the volumes are too
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double* ddef) {
    KP k;
    k.iVl = 1 / r[0];
    k.iVg = 1 / r[1];
    k.inv_m = 1 / r[2];
    k.KlaVl = r[3] * r[0];
    k.e = r[4];
    k.klap = k.iVl * k.KlaVl;
    k.eK = k.iVg * k.KlaVl;
    __builtin_memcpy(kp_out, &k, sizeof(k));
    ddef[0] = r[5];
    ddef[1] = r[6];
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0], u[1], u[2], u[3], u[0] * k.iVl, u[1] * k.iVg};
  }
  // sort key of the work-queue kernel: the faster of the two through-flow rates sets the stiffness, and with it the
  // number of (stability-limited) RK steps of an env step -- correlation with the measured step counts 0.83
  static constexpr bool COST_KEY = true;
  template <class K>
  PCG_DEV static double cost_key(const K& k, const double (&u)[NA + NDM]) {
    return __builtin_fmax(u[0] * k.iVl, u[1] * k.iVg);
  }
  // ---- hooks of the stiff integrator (PCG_INT_RODAS4, pcg_integrators.hpp) ----
  // End-point exponents of the two component groups at time-to-go tau (pcg_integrators.hpp "END-POINT ERROR
  // CONTROL"; twin: ep_exponents() in oracle/pcg_oracle.c).  Group 0 = liquid chain X (even components), group 1 = gas
  // chain Y.  A perturbation decays with the slower of the two through-flow rates a = L/Vl, c = G/Vg:
  //   k_s = trunc(min(kmax, ep_c min(a,c) tau)).
  // An error in the FAST chain dies at its own rate and only reaches the slow chain through the mass-transfer coupling,
  // attenuated by (coupling rate x residence time): kappa/c for Y -> X (kappa = Kla 2 Ymax / m with Ymax = 1, the top of
  // the observation box) and e/(a + Kla) for X -> Y (e = Kla Vl/Vg); with a safety factor 2:
  //   k_Y = min(trunc(min(kmax, ep_c c tau)), k_s + floor(log2(max(1, c/(2 kappa))))),  k_X likewise.
  // Measured on the action box of BASELINE configs[2] (tests/test_rodas4.py): the heaviest envs (low liquid flow, high
  // gas flow: the gas transient has to be resolved while the liquid barely forgets) drop from 136 to 102 attempts at
  // the same worst-case error; they are the critical path of a launch.
  static constexpr bool EP_GROUPS = true;
  PCG_DEV static constexpr int ep_group(int i) { return i & 1; }
  template <class K>
  PCG_DEV static void ep_exponents(const K& k, const double (&u)[NA + NDM], double ep_c, int kmax, double tau,
                                   int (&kg)[2]) {
#pragma clang fp contract(off)
    const double a = u[0] * k.iVl, c = u[1] * k.iVg;
    const double ct = ep_c * tau;
    const int ks = ep_trunc(ct * __builtin_fmin(a, c), kmax);
    if constexpr (!SQ) {  // the coupling bound below is the slope 2 Y / m of the eq_exponent == 2 curve: other exponents
      kg[0] = kg[1] = ks;  // keep the one exponent every component is entitled to
      return;
    }
    const double klap = k.iVl * k.KlaVl, e = k.iVg * k.KlaVl;
    const double kappa2 = (klap * 2.0 * k.inv_m) * 2.0, e2 = e * 2.0;  // x safety 2
    const int kX = ks + ep_ilog2((a + klap) / e2), kY = ks + ep_ilog2(c / kappa2);
    const int kXd = ep_trunc(ct * a, kmax), kYd = ep_trunc(ct * c, kmax);
    kg[0] = kX < kXd ? kX : kXd;
    kg[1] = kY < kYd ? kY : kYd;
  }
  // sort key of the work-queue kernel for the Rosenbrock pair: least-squares fit of its attempted steps per env step
  // over the action box (tools/ros4_costfit.py: correlation 0.92 with the ln d1 term the kernel adds); the expensive
  // lanes are the ones whose SLOW chain barely damps (small min rate), not the stiff ones
  template <class K>
  PCG_DEV static float cost_key_ros(const K& k, const double (&u)[NA + NDM]) {
    const float a = (float)(u[0] * k.iVl), c = (float)(u[1] * k.iVg);
    const float mn = __builtin_fminf(a, c), mx = __builtin_fmaxf(a, c);
    return 40.0f - 8.65f * __builtin_logf(mn) + 1.95f * __builtin_logf(mx) + 56.3f / mn;
  }
  // The cooperative rule (pcg_seulex.hpp): predicted attempts of the Rosenbrock pair for this env step, from the slower
  // through-flow rate mn and the scaled size d1 of f(x0) -- a least-squares fit over the action box of BASELINE configs[2]
  // (tools/prototypes/seulex8_calib.py: correlation 0.90 with the measured attempts; at threshold 48 it picks 7 %
  // of the envs, none below 25 attempts, and leaves none above 55; the default, 60, picks 2 %), in EXACT arithmetic: IEEE operations and exponent
  // extraction only, so that the kernels and the oracle pick the same envs.  Twin: me_coop_key() in oracle/pcg_oracle.c.
  template <class K>
  PCG_DEV static double coop_key(const K& k, const double (&u)[NA + NDM], double d1) {
#pragma clang fp contract(off)
    const double a = u[0] * k.iVl, c = u[1] * k.iVg;
    const double mn = __builtin_fmin(a, c);
    return ((-30.0 - 10.0 * plog2(mn)) + 3.6 / mn) + 4.0 * plog2(__builtin_fmax(d1, 1.0));
  }
  // W = theta I - J, analytic, eliminated in the natural order (X1,Y1,...,X5,Y5) without pivoting.  Rows:
  //   X_s:  DX X_s - alpha X_{s-1} - cx_s Y_s            alpha = L/Vl, Kla' = Kla, cx_s = Kla q_s, q_s = d(Y^e/m)/dY
  //   Y_s:  -e X_s + DY_s Y_s - beta Y_{s+1}             beta = G/Vg, e = Kla Vl/Vg, DY_s = theta + beta + e q_s
  // with DX = theta + alpha + Kla for every stage (the X pivots never change: fill-in only appears at (X_{s+1}, Y_s)).
  // For Y >= 0 the matrix is an M-matrix: no pivoting needed; a non-positive pivot (unphysical state) rejects the step.
  // Twin: me_ros_factor / me_ros_solve in oracle/pcg_oracle.c, operation for operation.
  static constexpr bool ROS_STRUCTURED = true;
  struct RosFac {
    double iDX, aD, eD, beta;
    double ct[5], iDY[5], m3[4];
    bool ok;
  };
  template <class K>
  PCG_DEV static void ros_factor(const K& k, const HoldT<double>& h, const double (&x)[NX], double theta, RosFac& F) {
#pragma clang fp contract(off)
    const double alpha = h.a, beta = h.c, klap = k.klap, e = k.eK;
    const double DX = theta + (alpha + klap);
    const double thb = theta + beta;
    F.iDX = rcp_ieee(DX);  // (== 1.0 / DX bit for bit: DX is finite and positive or the step is rejected)
    F.aD = alpha * F.iDX;
    F.eD = e * F.iDX;
    F.beta = beta;
    bool ok = (DX > 0.0) && (DX < __builtin_inf());
    double ct = 0.0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const double Y = x[2 * s + 1];
      double q;
      if constexpr (SQ) q = (2.0 * Y) * k.inv_m;
      else q = (k.e * pow(Y, k.e - 1.0)) * k.inv_m;
      const double cx = klap * q;
      ct = (s == 0) ? cx : __builtin_fma(F.m3[s > 0 ? s - 1 : 0], beta, cx);
      const double DY = __builtin_fma(e, q, thb);
      const double DYp = __builtin_fma(-F.eD, ct, DY);
      ok = ok && (DYp > 0.0) && (DYp < __builtin_inf());
      F.ct[s] = ct;
      F.iDY[s] = rcp_ieee(DYp);
      if (s < 4) F.m3[s] = (F.aD * ct) * F.iDY[s];
    }
    F.ok = ok;
  }
  // (a variant of both sweeps with ONE fused multiply-add per stage on the dependency chain -- 11 dependent operations
  // instead of 21, eight more multiplications off the chain -- measured no faster: the loop is issue-bound even for a wave
  // that has its SIMD to itself)
  PCG_DEV static void ros_solve(const RosFac& F, double (&b)[NX]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      b[2 * s + 1] = __builtin_fma(F.eD, b[2 * s], b[2 * s + 1]);
      if (s < 4) {
        b[2 * s + 2] = __builtin_fma(F.aD, b[2 * s], b[2 * s + 2]);
        b[2 * s + 2] = __builtin_fma(F.m3[s], b[2 * s + 1], b[2 * s + 2]);
      }
    }
#pragma unroll
    for (int ss = 0; ss < 5; ++ss) {
      const int s = 4 - ss;
      const double yv = ((s == 4) ? b[2 * s + 1] : __builtin_fma(F.beta, b[s < 4 ? 2 * s + 3 : 0], b[2 * s + 1])) * F.iDY[s];
      b[2 * s + 1] = yv;
      b[2 * s] = __builtin_fma(F.ct[s], yv, b[2 * s]) * F.iDX;
    }
  }
  // With eq_exponent == 2 the right-hand side is an exactly specified sequence of IEEE operations (contraction off,
  // fused multiply-adds where written): this model runs the adaptive pair at its stability limit, where the
  // step-size sequence amplifies a last-bit difference (tests/helpers.py) -- with a fixed operation order every
  // kernel, and the oracle's twin of this function, produce the same bits.
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const R X = x[2 * s], Y = x[2 * s + 1];
      const R Xp = (s == 0) ? h.X0 : x[2 * s - 2];
      const R Yn = (s == 4) ? h.Y6 : x[2 * s + 3];
      if constexpr (SQ) {
        // the reference's expressions (model_classes.py:370-412) with the constant factors folded (round 5): per stage
        //   q = X - Y^2 / m,   dX = (L / Vl) (X' - X) - Kla q,   dY = (G / Vg) (Y" - Y) + (Kla Vl / Vg) q
        // 8 fp64 operations against 9 (Q = Kla Vl q, then 1 / Vl and 1 / Vg over each sum): the explicit pair's attempt is
        // 7 of these evaluations.  Twin: rhs_me_kernel_order() in oracle/pcg_oracle.c, operation for operation.
#ifdef PCG_ME_RHS_UNFOLDED  // the form of rounds 1-4 (A/B builds: tools/fastlib.sh ... -DPCG_ME_RHS_UNFOLDED; not the oracle's twin)
        const R Q = k.KlaVl * pk_fma(-(Y * Y), k.inv_m, X);
        dx[2 * s] = k.iVl * pk_fma(h.L, Xp - X, -Q);
        dx[2 * s + 1] = k.iVg * pk_fma(h.G, Yn - Y, Q);
#else
        const R q = pk_fma(-(Y * Y), k.inv_m, X);
        dx[2 * s] = pk_fma(h.a, Xp - X, -(k.klap * q));
        dx[2 * s + 1] = pk_fma(h.c, Yn - Y, k.eK * q);
#endif
      } else {
        const R Q = k.KlaVl * (X - eq_curve<SQ>(Y, k.e, k.inv_m));
        dx[2 * s] = k.iVl * (h.L * (Xp - X) - Q);
        dx[2 * s + 1] = k.iVg * (h.G * (Yn - Y) + Q);
      }
    }
  }
};
template <>
struct Model<PCG_MODEL_ME> : MEImpl<false> {};
template <>
struct Model<PCG_KID_ME_SQ> : MEImpl<true> {};


// ---------------------------------------------------------------------------
// multistage_extraction_reactive -- model_classes.py:763-861.
// raw = Vl,Vg,m,Kla,k,eq_exponent,XA0,YA6,YB6,YC6 ; x = (XA,YA,YB,YC) x 5
// ---------------------------------------------------------------------------
template <bool SQ>
struct MEReactiveImpl {
  static constexpr int NX = 20, NA = 2, NDM = 0, NRAW = 10;
  static constexpr bool DYNAMIC = false;
  static constexpr bool FULL = true;  // all kernel specialisations
  struct KP {
    double iVl, iVg, inv_m, KlaVl, kVg, e, XA0, YA6, YB6, YC6;
    double Kla, eK, kr;  // folded for rhs(): Kla = KlaVl / Vl, eK = KlaVl / Vg, kr = k
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R l, g;  // through-flow rates L / Vl, G / Vg: held over the env step, so the divisions by the volumes are too
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k;
    k.iVl = 1 / r[0];
    k.iVg = 1 / r[1];
    k.inv_m = 1 / r[2];
    k.KlaVl = r[3] * r[0];
    k.kVg = r[4] * r[1];
    k.e = r[5];
    k.XA0 = r[6];
    k.YA6 = r[7];
    k.YB6 = r[8];
    k.YC6 = r[9];
    k.Kla = r[3];
    k.eK = r[3] * r[0] / r[1];
    k.kr = r[4];
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0] * k.iVl, u[1] * k.iVg};
  }
  static constexpr bool COST_KEY = true;  // as for the 10-state model
  template <class K>
  PCG_DEV static double cost_key(const K& k, const double (&u)[NA + NDM]) {
    return __builtin_fmax(u[0] * k.iVl, u[1] * k.iVg);
  }
  // end-point error control of PCG_INT_RODAS4 (dense linear algebra here): one exponent for every component, from the
  // slower through-flow rate (see MEImpl::ep_exponents)
  static constexpr bool EP_GROUPS = true;
  PCG_DEV static constexpr int ep_group(int) { return 0; }
  template <class K>
  PCG_DEV static void ep_exponents(const K& k, const double (&u)[NA + NDM], double ep_c, int kmax, double tau,
                                   int (&kg)[2]) {
#pragma clang fp contract(off)
    const double ct = ep_c * tau;
    kg[0] = kg[1] = ep_trunc(ct * __builtin_fmin(u[0] * k.iVl, u[1] * k.iVg), kmax);
  }
  // The reference's expressions (model_classes.py:790-845) with the constant factors folded: per stage
  //   q = XA - YA^e / m,  r = k YA YB
  //   dXA = l (XA' - XA) - Kla q        dYA = g (YA" - YA) + (Kla Vl / Vg) q - r
  //   dYB = g (YB" - YB) - r            dYC = g (YC" - YC) + r
  // 14 fp64 operations against 18 as written there (Q = Kla Vl q, r Vg, then 1/Vl and 1/Vg over each sum): the explicit
  // pair's attempt is ~540 of its ~1300 operations in this function.  Same values to rounding (RHS parity <= 1e-12).
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const R XA = x[4 * s], YA = x[4 * s + 1], YB = x[4 * s + 2], YC = x[4 * s + 3];
      const R q = XA - eq_curve<SQ>(YA, k.e, k.inv_m);
      const R r = (k.kr * YA) * YB;
      const R XAp = (s == 0) ? R(k.XA0) : x[4 * s - 4];
      const R YAn = (s == 4) ? R(k.YA6) : x[4 * s + 5];
      const R YBn = (s == 4) ? R(k.YB6) : x[4 * s + 6];
      const R YCn = (s == 4) ? R(k.YC6) : x[4 * s + 7];
      dx[4 * s + 0] = h.l * (XAp - XA) - k.Kla * q;
      dx[4 * s + 1] = h.g * (YAn - YA) + (k.eK * q - r);
      dx[4 * s + 2] = h.g * (YBn - YB) - r;
      dx[4 * s + 3] = h.g * (YCn - YC) + r;
    }
  }
};
template <>
struct Model<PCG_MODEL_ME_REACTIVE> : MEReactiveImpl<false> {};
template <>
struct Model<PCG_KID_ME_REACTIVE_SQ> : MEReactiveImpl<true> {};


// ---------------------------------------------------------------------------
// crystallization -- model_classes.py:1232-1345.
// raw = ka,kb,kc,kd,kg,k1,k2,a,b,alfa,ro ; x = mu0..mu3,conc,CV,Ln ; u = [T degC]
// The input-only factors (Ceq(T), exp(kb/Tk), exp(k1/Tk)) are constant over the
// held step and are evaluated once per env step, not once per RK stage.
// ---------------------------------------------------------------------------
template <>
struct Model<PCG_MODEL_CRYST> {
  static constexpr int NX = 7, NA = 1, NDM = 0, NRAW = 11;
  static constexpr bool DYNAMIC = false;
  static constexpr bool FULL = true;  // all kernel specialisations
  struct KP {
    double ka, kb, kc2, kd2, kg, k1, k22, a, b, cc;  // kc2 = kc/2 ..., cc = -0.5*ro*alfa
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R Ceq, eB, eG;  // eB = ka*exp(kb/Tk), eG = kg*exp(k1/Tk)
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k;
    k.ka = r[0];
    k.kb = r[1];
    k.kc2 = r[2] / 2;
    k.kd2 = r[3] / 2;
    k.kg = r[4];
    k.k1 = r[5];
    k.k22 = r[6] / 2;
    k.a = r[7];
    k.b = r[8];
    k.cc = -0.5 * r[10] * r[9];
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    const R Tk = u[0] + 273.15;
    HoldT<R> h;
    h.Ceq = -686.2686 + 3.579165 * Tk - 0.00292874 * (Tk * Tk);
    h.eB = k.ka * exp(k.kb / Tk);
    h.eG = k.kg * exp(k.k1 / Tk);
    return h;
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R mu0 = x[0], mu1 = x[1], mu2 = x[2], mu3 = x[3], conc = x[4];
    const R S = conc * 1e3 - h.Ceq;
    const R S2 = S * S;
    // (S^2)^(kc/2) (mu3^2)^(kd/2) and (S^2)^(k2/2) share log(S^2): two logs + two exps instead of the
    // reference's three pow() (model_classes.py:1299-1300) -- a third of the instructions; the exponent
    // is O(30), so the result is within ~1e-14 relative of the pow form (parity bar 1e-12).  log_pos / exp_bounded:
    // range-restricted forms (pcg_pack.hpp), S^2 and mu3^2 are positive and far from the denormal range.
    const R L1 = log_pos(S2), L3 = log_pos(mu3 * mu3);
    const R B0 = h.eB * exp_bounded(k.kc2 * L1 + k.kd2 * L3);
    const R Ginf = h.eG * exp_bounded(k.k22 * L1);
    const R m12 = k.a * mu1 * 1e-4 + k.b * mu2 * 1e-8;
    const R m23 = k.a * mu2 * 1e-8 + k.b * mu3 * 1e-12;
    const R d0 = B0;
    const R d1 = Ginf * (k.a * mu0 + k.b * mu1 * 1e-4) * 1e4;
    const R d2 = 2.0 * Ginf * m12 * 1e8;
    const R d3 = 3.0 * Ginf * m23 * 1e12;
    const R mu1sq = mu1 * mu1;
    const R CV = sqrt_pos(div_fast(mu2 * mu0, mu1sq) - 1.0);
    dx[0] = d0;
    dx[1] = d1;
    dx[2] = d2;
    dx[3] = d3;
    dx[4] = k.cc * Ginf * m23;
    // one divide for the two denominators of dCV/dt (model_classes.py:1314), Newton divides throughout
    dx[5] = div_fast((d2 * mu0 + mu2 * d0) * mu1sq - mu2 * mu0 * 2.0 * mu1 * d1,
                     (2.0 * CV + 1e-10) * (mu1sq * mu1sq + 1e-10));
    dx[6] = div_fast(d1 * mu0 - mu1 * d0, mu0 * mu0 + 1e-10);
  }
};

// ---------------------------------------------------------------------------
// affine custom model -- dx = A x + B u + c  (pcgym.py:150-153 custom_model whose
// RHS is affine; the reference's only KAT, tests/environment/
// test_make_env_custom_model.py:66-86).  raw = A[nx][nx] | B[nx][nu] | c[nx].
// Padded with zeros to 8 states / 4 inputs: padded states have dx = 0.
// ---------------------------------------------------------------------------
template <>
struct Model<PCG_MODEL_AFFINE> {
  static constexpr int NX = 8, NA = 4, NDM = 0, NRAW = -1;
  static constexpr bool DYNAMIC = true;  // runtime nx<=8, na<=4
  static constexpr bool FULL = true;
  struct KP {
    double A[8][8], Bm[8][4], c[8];
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R f[8];  // B u + c
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int nx, int nu, double* kp_out, double*) {
    KP k;
    __builtin_memset(&k, 0, sizeof(k));
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) k.A[i][j] = r[i * nx + j];
      for (int j = 0; j < nu; ++j) k.Bm[i][j] = r[nx * nx + i * nu + j];
      k.c[i] = r[nx * nx + nx * nu + i];
    }
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    HoldT<R> h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      R s = k.c[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) s = s + k.Bm[i][j] * u[j];
      h.f[i] = s;
    }
    return h;
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      R s = h.f[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s = s + k.A[i][j] * x[j];
      dx[i] = s;
    }
  }
};

// ===========================================================================
// "next" row f-2: further registry models of pcgym.py:128-148.  Same functor
// template; FULL = false instantiates the general kernels only (no streaming / pipelined / LDS-stage
// specialisations), which keeps the build time of the library bounded.
// ===========================================================================

// complex_cstr -- model_classes.py:65-125.  raw = q,V,rho,C,deltaHr1,EA1_over_R,k01,deltaHr2,EA2_over_R,k02,UA,Ti,Caf
template <>
struct Model<PCG_MODEL_COMPLEX_CSTR> {
  static constexpr int NX = 4, NA = 1, NDM = 2, NRAW = 13;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double qV, k01, nEA1, k02, nEA2, h1, h2, c2;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R Tc, Ti, Caf;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double* ddef) {
    KP k;
    k.qV = r[0] / r[1];
    k.k01 = r[6];
    k.nEA1 = -r[5];
    k.k02 = r[9];
    k.nEA2 = -r[8];
    k.h1 = (-r[4]) / (r[2] * r[3]);
    k.h2 = (-r[7]) / (r[2] * r[3]);
    k.c2 = r[10] / (r[2] * r[3] * r[1]);
    __builtin_memcpy(kp_out, &k, sizeof(k));
    ddef[0] = r[11];
    ddef[1] = r[12];
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0], u[1], u[2]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R ca = x[0], cb = x[1], cc = x[2], T = x[3];
    const R r1 = k.k01 * exp_bounded(div_fast(k.nEA1, T)) * ca;
    const R r2 = k.k02 * exp_bounded(div_fast(k.nEA2, T)) * cb;
    dx[0] = k.qV * (h.Caf - ca) - r1;
    dx[1] = k.qV * (0.0 - cb) + 2.0 * r1 - r2;
    dx[2] = k.qV * (0.0 - cc) + r2;
    dx[3] = k.qV * (h.Ti - T) + (k.h1 * r1 + k.h2 * r2) + k.c2 * (h.Tc - T);
  }
};

// disease_model (SIRS with vaccination) -- model_classes.py:156-183.  raw = beta, gamma
template <>
struct Model<PCG_MODEL_DISEASE> {
  static constexpr int NX = 3, NA = 1, NDM = 0, NRAW = 2;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double beta, gamma;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R u;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{r[0], r[1]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R S = x[0], I = x[1];
    const R inf = k.beta * S * I;
    dx[0] = -inf - h.u * S;
    dx[1] = inf - k.gamma * I;
    dx[2] = k.gamma * I + h.u * S;
  }
};

// batch (exothermic consecutive reactions) -- model_classes.py:222-265.
// raw = k01,k02,EA1,EA2,R,dH1,dH2,rho,Cp,UA,V
template <>
struct Model<PCG_MODEL_BATCH> {
  static constexpr int NX = 4, NA = 1, NDM = 0, NRAW = 11;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double k01, k02, nE1, nE2, g1, g2, c;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R Tc;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k;
    k.k01 = r[0];
    k.k02 = r[1];
    k.nE1 = -r[2] / r[4];
    k.nE2 = -r[3] / r[4];
    k.g1 = r[5] / (r[7] * r[8]);
    k.g2 = r[6] / (r[7] * r[8]);
    k.c = r[9] / (r[7] * r[8] * r[10]);
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R CA = x[0], CB = x[1], T = x[3];
    const R r1 = k.k01 * exp(k.nE1 / T) * CA;
    const R r2 = k.k02 * exp(k.nE2 / T) * CB;
    dx[0] = -r1;
    dx[1] = 2.0 * r1 - r2;
    dx[2] = r2;
    dx[3] = -(k.g1 * r1 + k.g2 * r2) + k.c * (h.Tc - T);
  }
};

// photo_production ("photobioreactor") -- model_classes.py:433-506.
// raw = u_m,u_d,Y_NX,k_m,k_d,k_sq,K_Nq,k_iq,k_s,k_i,k_N ; u = [I, F_N]
// the light-dependent factors only depend on the held input: once per env step
template <>
struct Model<PCG_MODEL_PHOTO> {
  static constexpr int NX = 3, NA = 2, NDM = 0, NRAW = 11;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double u_m, u_d, Y_NX, k_m, k_d, k_sq, K_Nq, k_iq, k_s, k_i, k_N;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R f1, f2, F_N;  // u_m I/(I+k_s+I^2/k_i),  k_m I/(I+k_sq+I^2/k_iq)
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    const R I = u[0];
    HoldT<R> h;
    h.f1 = k.u_m * I / (I + k.k_s + (I * I) / k.k_i);
    h.f2 = k.k_m * I / (I + k.k_sq + (I * I) / k.k_iq);
    h.F_N = u[1];
    return h;
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R c_x = x[0], c_N = x[1], c_q = x[2];
    const R g = h.f1 * c_x * c_N / (c_N + k.k_N);
    dx[0] = g - k.u_d * c_x;
    dx[1] = -k.Y_NX * g + h.F_N;
    dx[2] = h.f2 * c_x - (k.k_d * c_q) / (c_N + k.K_Nq);
  }
};

// cstr_series_recycle -- model_classes.py:611-679.  raw = C_O,T_O,V1,V2,U1A1,U2A2,rho,cp,k,E,deltaH,R
// u = [F, L, Tc1, Tc2]
template <>
struct Model<PCG_MODEL_CSTR_SERIES> {
  static constexpr int NX = 4, NA = 4, NDM = 0, NRAW = 12;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double COV1, TOV1, iV1, iV2, c1, c2, k, nER, kh;  // c_i = U_iA_i/(V_i rho cp), kh = k(-dH)/(rho cp)
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R F, L, Tc1, Tc2;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k;
    k.COV1 = r[0] / r[2];
    k.TOV1 = r[1] / r[2];
    k.iV1 = 1 / r[2];
    k.iV2 = 1 / r[3];
    k.c1 = r[4] / (r[2] * r[6] * r[7]);
    k.c2 = r[5] / (r[3] * r[6] * r[7]);
    k.k = r[8];
    k.nER = -r[9] / r[11];
    k.kh = (r[8] * (-r[10])) / (r[6] * r[7]);
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0], u[1], u[2], u[3]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R C1 = x[0], T1 = x[1], C2 = x[2], T2 = x[3];
    const R e1 = exp(k.nER / T1), e2 = exp(k.nER / T2);
    const R FL = h.F + h.L;
    dx[0] = k.COV1 * h.F + k.iV1 * h.L * C2 - k.iV1 * FL * C1 - k.k * C1 * e1;
    dx[1] = k.TOV1 * h.F + k.iV1 * h.L * T2 - k.c1 * (T1 - h.Tc1) - k.iV1 * FL * T1 + k.kh * C1 * e1;
    dx[2] = k.iV2 * FL * (C1 - C2) - k.k * C2 * e2;
    dx[3] = k.iV2 * FL * (T1 - T2) - k.c2 * (T2 - h.Tc2) + k.kh * C2 * e2;
  }
};

// distillation_column -- model_classes.py:682-760.  raw = D,q,alpha,X_feed,M0,Mb,M ; u = [R, F]
// x = X0,X1,X2,X3,Xf,X4,X5,X6,Xb
template <>
struct Model<PCG_MODEL_DISTILLATION> {
  static constexpr int NX = 9, NA = 2, NDM = 0, NRAW = 7;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double D, q, alpha, am1, X_feed, iM0, iMb, iM;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R L, V, Ld, Vd, W, FXf;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{r[0], r[1], r[2], r[2] - 1, r[3], 1 / r[4], 1 / r[5], 1 / r[6]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    const R Rr = u[0], F = u[1];
    HoldT<R> h;
    h.L = Rr * k.D;
    h.V = (Rr + 1.0) * k.D;
    h.Ld = h.L + k.q * F;
    h.Vd = h.V + (1 - k.q) * F;
    h.W = F - k.D;
    h.FXf = F * k.X_feed;
    return h;
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    R Y[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) Y[i] = (k.alpha * x[i]) / (1.0 + k.am1 * x[i]);
    // indices: 0 X0, 1 X1, 2 X2, 3 X3, 4 Xf, 5 X4, 6 X5, 7 X6, 8 Xb
    dx[0] = k.iM0 * ((h.V * Y[1]) - (h.L + k.D) * x[0]);
    dx[1] = k.iM * (h.L * (x[0] - x[1]) + h.V * (Y[2] - Y[1]));
    dx[2] = k.iM * (h.L * (x[1] - x[2]) + h.V * (Y[3] - Y[2]));
    dx[3] = k.iM * (h.L * (x[2] - x[3]) + h.V * (Y[4] - Y[3]));
    dx[4] = k.iM * (h.L * x[3] - h.Ld * x[4] + h.Vd * Y[5] - h.V * Y[4] + h.FXf);
    dx[5] = k.iM * (h.Ld * (x[4] - x[5]) + h.Vd * (Y[6] - Y[5]));
    dx[6] = k.iM * (h.Ld * (x[5] - x[6]) + h.Vd * (Y[7] - Y[6]));
    dx[7] = k.iM * (h.Ld * (x[6] - x[7]) + h.Vd * (Y[8] - Y[7]));
    dx[8] = k.iMb * (h.Ld * x[7] - h.W * x[8] - h.Vd * Y[8]);
  }
};

// polymerisation_reactor -- model_classes.py:1158-1229.
// raw = Ap,Ad,At,Ep_over_R,Ed_over_R,Et_over_R,f,V,deltaHp,rho,cp ; x = T,M,I ; u = [F,Tf,Mf,If]
template <>
struct Model<PCG_MODEL_POLYMER> {
  static constexpr int NX = 3, NA = 4, NDM = 0, NRAW = 11;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double Ap, Ad, At, nEp, nEd, nEt, f, hq, iV;  // hq = (-deltaHp)/(rho cp)
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R FV, Tf, Mf, If;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{r[0], r[1], r[2], -r[3], -r[4], -r[5], r[6], (-r[8]) / (r[9] * r[10]), 1 / r[7]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0] * k.iV, u[1], u[2], u[3]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
    const R T = x[0], M = x[1], I = x[2];
    const R kp = k.Ap * exp(k.nEp / T), kd = k.Ad * exp(k.nEd / T), kt = k.At * exp(k.nEt / T);
    const R ri = 2.0 * k.f * kd * I;
    const R rp = kp * sqrt((k.f * kd * I) / kt);
    dx[0] = h.FV * (h.Tf - T) + k.hq * rp;
    dx[1] = h.FV * (h.Mf - M) - rp;
    dx[2] = h.FV * (h.If - I) - ri;
  }
};

// biofilm_reactor -- model_classes.py:1046-1155.
// raw = V,Va,Kla,m,eq_exponent,O_air,vm_1,vm_2,K1,K2,KO_1,KO_2 ; u = [F,Fr,S1_F,S2_F,S3_F]
// x = (S1,S2,S3,O) for reactor stages 1..3, then the absorber tank A
template <>
struct Model<PCG_MODEL_BIOFILM> {
  static constexpr int NX = 16, NA = 5, NDM = 0, NRAW = 12;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double iV, iVa, Kla, OAeq, vm1, vm2, K1, K2, KO1, KO2;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R FrV, FrVa, FVa, S1F, S2F, S3F;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{1 / r[0], 1 / r[1], r[2], pow(r[5], r[4]) / r[3], r[6], r[7], r[8], r[9], r[10], r[11]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[1] * k.iV, u[1] * k.iVa, u[0] * k.iVa, u[2], u[3], u[4]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int c = 4 * s, pv = (s == 0) ? 12 : 4 * (s - 1);  // upstream: absorber for stage 1
      const R S1 = x[c], S2 = x[c + 1], S3 = x[c + 2], O = x[c + 3];
      const R r1 = ((k.vm1 * S1) / (k.K1 + S1)) * (O / (k.KO1 + O));
      const R r2 = ((k.vm2 * S2) / (k.K2 + S2)) * (O / (k.KO2 + O));
      // the reference subtracts rs1 = -r1, rs2 = r1 - r2, rs3 = r2, ro = -3.5 r1 - 1.1 r2 (signs kept as written there)
      dx[c] = h.FrV * (x[pv] - S1) + r1;
      dx[c + 1] = h.FrV * (x[pv + 1] - S2) - (r1 - r2);
      dx[c + 2] = h.FrV * (x[pv + 2] - S3) - r2;
      dx[c + 3] = h.FrV * (x[pv + 3] - O) - (-r1 * 3.5 - r2 * 1.1);
    }
    dx[12] = h.FrVa * (x[8] - x[12]) + h.FVa * (h.S1F - x[12]);
    dx[13] = h.FrVa * (x[9] - x[13]) + h.FVa * (h.S2F - x[13]);
    dx[14] = h.FrVa * (x[10] - x[14]) + h.FVa * (h.S3F - x[14]);
    dx[15] = h.FrVa * (x[11] - x[15]) + k.Kla * (k.OAeq - x[15]);
  }
};

// heat_exchanger -- model_classes.py:935-1044.  raw = Utm,Usm,L,Dt,Dm,Ds,cpt,cpm,cps,rhot,rhom,rhos
// x = (Tt,Tm,Ts) for segments 1..8 ; u = [Ft,Fs,Tt0,Ts9]  (tube flows 1->8, shell flows 8->1)
template <>
struct Model<PCG_MODEL_HEAT_EX> {
  static constexpr int NX = 24, NA = 4, NDM = 0, NRAW = 12;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double ct, cm, cs, UAt, UAm, cpt, cps;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R Ftc, Fsc, Tt0, Ts9;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    const double pi = 3.141592653589793;
    const double L = r[2], Dt = r[3], Dm = r[4], Ds = r[5];
    const double Vt = L * pi * (Dt * Dt), At = L * pi * Dt, Vm = L * pi * (Dm * Dm - Dt * Dt), Am = L * pi * Dm;
    const double Vs = L * pi * (Ds * Ds - Dm * Dm);
    KP k{1 / (r[6] * r[9] * Vt), 1 / (r[7] * r[10] * Vm), 1 / (r[8] * r[11] * Vs), r[0] * At, r[1] * Am, r[6], r[8]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K& k, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0] * k.cpt, u[1] * k.cps, u[2], u[3]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>& h, const R (&x)[NX], R (&dx)[NX]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const R Tt = x[3 * i], Tm = x[3 * i + 1], Ts = x[3 * i + 2];
      const R Qt = k.UAt * (Tt - Tm), Qm = k.UAm * (Tm - Ts);
      const R Tt_up = (i == 0) ? h.Tt0 : x[3 * (i - 1)];
      const R Ts_up = (i == 7) ? h.Ts9 : x[3 * (i + 1) + 2];
      dx[3 * i] = k.ct * (h.Ftc * (Tt_up - Tt) - Qt);
      dx[3 * i + 1] = k.cm * (Qt - Qm);
      dx[3 * i + 2] = k.cs * (h.Fsc * (Ts_up - Ts) + Qm);
    }
  }
};

// invariant_batch -- model_classes.py:268-293.  raw = k1f,k1r,k2f,k2r ; x = xA,xB,xC,xD ; no inputs:
// the kernels carry one dummy action that the RHS ignores (the host passes zeros).
template <>
struct Model<PCG_MODEL_INV_BATCH> {
  static constexpr int NX = 4, NA = 1, NDM = 0, NRAW = 4;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double k1f, k1r, k2f, k2r;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R none;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{r[0], r[1], r[2], r[3]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>&, const R (&x)[NX], R (&dx)[NX]) {
    const R r1 = k.k1f * x[0] * x[1] - k.k1r * x[2];
    const R r2 = k.k2f * x[0] * x[2] - k.k2r * x[3];
    dx[0] = -r1 - r2;
    dx[1] = -r1;
    dx[2] = r1 - r2;
    dx[3] = r2;
  }
};

// coupled_oscillators -- model_classes.py:186-216, N = 10 (ring of masses).  raw = N,k,m ;
// x = x1..x10, p1..p10 ; no inputs (dummy action as above).
template <>
struct Model<PCG_MODEL_OSCILLATORS> {
  static constexpr int NX = 20, NA = 1, NDM = 0, NRAW = 3;
  static constexpr bool DYNAMIC = false, FULL = false;
  struct KP {
    double k, m;
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R none;
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double* r, int, int, double* kp_out, double*) {
    KP k{r[1], r[2]};
    __builtin_memcpy(kp_out, &k, sizeof(k));
  }
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    return HoldT<R>{u[0]};
  }
  template <class R, class K>
  PCG_DEV static void rhs(const K& k, const HoldT<R>&, const R (&x)[NX], R (&dx)[NX]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      dx[i] = x[10 + i] / k.m;
      dx[10 + i] = -k.k * (2.0 * x[i] - x[(i + 9) % 10] - x[(i + 1) % 10]);
    }
  }
};

// ---------------------------------------------------------------------------
// PCG_MODEL_USER -- custom_model with an arbitrary right-hand side (pcgym.py:150-153: any object with
// __call__(x, u) and info()).  Exists only inside a run-time compiled translation unit: pcg_abi.hip defines the sizes
// as macros, splices the user's statements into pcg_user_rhs() and instantiates the general kernels for this model.
// Parameters live in the 136-double constant block the affine custom model uses (DevConst::kp_big).
// ---------------------------------------------------------------------------
#ifdef PCG_USER_NX
PCG_DEV void pcg_user_rhs(const double* x, const double* u, const double* p, double* dx);
template <>
struct Model<PCG_MODEL_USER> {
  static constexpr int NX = PCG_USER_NX, NA = PCG_USER_NA, NDM = PCG_USER_NDM, NRAW = PCG_USER_NP;
  static constexpr bool DYNAMIC = false;
  static constexpr bool FULL = false;
  static constexpr bool KP_BIG = true;
  struct KP {
    double p[NRAW > 0 ? NRAW : 1];
  };
  using CKP = const PCG_CONSTANT KP;
  template <class R>
  struct HoldT {
    R u[NA + NDM];
  };
  using Hold = HoldT<double>;
  PCG_HD static void prep(const double*, int, int, double*, double*) {}
  template <class R, class K>
  PCG_DEV static HoldT<R> hold(const K&, const R (&u)[NA + NDM]) {
    HoldT<R> h;
#pragma unroll
    for (int i = 0; i < NA + NDM; ++i) h.u[i] = u[i];
    return h;
  }
  template <class K>
  PCG_DEV static void rhs(const K& k, const HoldT<double>& h, const double (&x)[NX], double (&dx)[NX]) {
    double p[NRAW > 0 ? NRAW : 1];
#pragma unroll
    for (int i = 0; i < NRAW; ++i) p[i] = k.p[i];  // scalar loads from the constant block
    pcg_user_rhs(x, h.u, p, dx);
  }
};
#endif

}  // namespace pcg
