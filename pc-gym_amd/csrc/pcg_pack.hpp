// pcg_pack.hpp -- W environments per lane as one value type.
//
// A wave that integrates ONE env per lane executes a single chain of dependent fp64 instructions
// (RK stage -> exp -> divide -> ...), and a dependent fp64 op can only issue every ~8-10 cycles:
// tools/overlapbench.hip measures 36.7 us for 256 dependent FMAs per env against 19.0 us when the
// same work is two independent chains.  Pack<W> carries W envs through the same code, so every
// operator becomes W independent instructions the scheduler interleaves: the VALU issues
// back-to-back, a wave finishes its arithmetic sooner, and memory accesses are 8*W bytes per lane.
//
// Pack<1> is a plain double in a struct: one code path for both widths.
#pragma once
#ifndef __HIPCC_RTC__  // built in under hipRTC
#include <hip/hip_runtime.h>
#endif

namespace pcg {

// the handful of type traits the device code needs, spelled out here so that the kernel headers also compile under
// hipRTC (run-time compilation of user expressions, pcgym_amd/jit.py), which ships no <type_traits>
namespace tt {
struct true_type { static constexpr bool value = true; };
struct false_type { static constexpr bool value = false; };
template <class A, class B> struct is_same : false_type {};
template <class A> struct is_same<A, A> : true_type {};
template <bool C, class A, class B> struct conditional { using type = A; };
template <class A, class B> struct conditional<false, A, B> { using type = B; };
template <class...> using void_t = void;
}  // namespace tt

#define PCG_PK __device__ __forceinline__

template <int W>
struct Pack {
  double v[W];
  PCG_PK Pack() {}
  PCG_PK Pack(double s) {
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = s;
  }
};

#define PCG_PACK_BINOP(OP)                                                  \
  template <int W>                                                          \
  PCG_PK Pack<W> operator OP(const Pack<W>& a, const Pack<W>& b) {          \
    Pack<W> r;                                                              \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = a.v[i] OP b.v[i]; \
    return r;                                                               \
  }                                                                         \
  template <int W>                                                          \
  PCG_PK Pack<W> operator OP(const Pack<W>& a, double b) {                  \
    Pack<W> r;                                                              \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = a.v[i] OP b;     \
    return r;                                                               \
  }                                                                         \
  template <int W>                                                          \
  PCG_PK Pack<W> operator OP(double a, const Pack<W>& b) {                  \
    Pack<W> r;                                                              \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = a OP b.v[i];     \
    return r;                                                               \
  }
PCG_PACK_BINOP(+)
PCG_PACK_BINOP(-)
PCG_PACK_BINOP(*)
PCG_PACK_BINOP(/)
#undef PCG_PACK_BINOP

template <int W>
PCG_PK Pack<W> operator-(const Pack<W>& a) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = -a.v[i];
  return r;
}
template <int W>
PCG_PK Pack<W>& operator+=(Pack<W>& a, const Pack<W>& b) {
#pragma unroll
  for (int i = 0; i < W; ++i) a.v[i] += b.v[i];
  return a;
}

#define PCG_PACK_FN1(NAME)                                           \
  template <int W>                                                   \
  PCG_PK Pack<W> NAME(const Pack<W>& a) {                            \
    Pack<W> r;                                                       \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = ::NAME(a.v[i]); \
    return r;                                                        \
  }
PCG_PACK_FN1(exp)
PCG_PACK_FN1(log)
PCG_PACK_FN1(sqrt)
PCG_PACK_FN1(fabs)
#undef PCG_PACK_FN1

// explicit fused multiply-add for both value types (model code whose rounding is part of a bit-exact contract)
PCG_PK double pk_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <int W>
PCG_PK Pack<W> pk_fma(const Pack<W>& a, const Pack<W>& b, const Pack<W>& c) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = __builtin_fma(a.v[i], b.v[i], c.v[i]);
  return r;
}
template <int W>
PCG_PK Pack<W> pk_fma(const Pack<W>& a, double b, const Pack<W>& c) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = __builtin_fma(a.v[i], b, c.v[i]);
  return r;
}

// ---- fast fp64 helpers for arguments of known range (no special-case handling) -------------------
// 1/x: hardware estimate + two Newton steps, ~1 ulp; x finite, normal, non-zero.  6 VALU instructions
// against 11 for the IEEE divide sequence (div_scale x2, rcp, 5 fma, div_fmas, div_fixup).
PCG_PK double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
// 1/x CORRECTLY ROUNDED for a finite, normal x whose reciprocal is normal too: the arithmetic core of the compiler's IEEE
// division (estimate, two Newton steps, quotient residual, final fused correction) without the operand scaling
// (v_div_scale x2) and the special-case fix-up (v_div_fixup) that bracket it -- 6 dependent instructions instead of 10.
// Bit-identical to `1.0 / x` on such arguments (the bit-exact Rodas4 parity tests against the oracle's `1.0 / x` hold it
// to that); zero, infinite or NaN arguments give NaN where the division gives inf / 0 (callers reject those steps anyway).
PCG_PK double rcp_ieee(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// a/b to ~0.5 ulp: hardware estimate, ONE Newton step, then one residual correction of the quotient (6 instructions).
// v_rcp_f64 is good to 2^-24.4 (tools/issuebench.hip, 4M arguments), so r carries 2^-48.8 after one step and the corrected
// quotient q + r (a - b q) an error of (2^-48.8)^2 relative -- below its own rounding: the second Newton step of rounds
// 1-3 bought nothing.  (rcp_fast() keeps two: its result is used as it is.)
PCG_PK double div_fast(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  double q = a * r;
  q = __builtin_fma(__builtin_fma(-b, q, a), r, q);
  return q;
}
// exp(x) for x in [-745, 700] (clamped below; NaN propagates): Cody-Waite reduction by ln2, then 1 + r + r^2 q(r) on
// |r| <= ln2/2 with q of degree 9: the minimax fit of tools/prototypes/exp_minimax.py (truncation 9e-18 relative with these
// doubles; round 1-3 used the degree-13 Taylor polynomial, 4e-18: two more multiply-adds for nothing), ~1 ulp like the
// library's.  18 VALU instructions against ~30 for the library exp(), whose extra work is overflow / underflow / NaN
// selection.  The lower clamp replaces only the HIGH word of an argument below -745 (one compare + one 32-bit select
// instead of a 64-bit select: the low word of such an argument moves -745 by < 5e-4, the result is the smallest
// denormal or 0 either way), and a NaN fails the compare and stays NaN (fmax would return -745).
PCG_PK double exp_bounded(double x) {
#ifdef PCG_EXP_TAYLOR13  // the polynomial and clamp of rounds 1-3 (A/B builds)
  x = (x < -745.0) ? -745.0 : x;
#else
  {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned hi = (x < -745.0) ? 0xC0874800u : (unsigned)(b >> 32);
    x = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (b & 0xFFFFFFFFull)));
  }
#endif
  const double n = __builtin_rint(x * 1.44269504088896338700e+00);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
#ifdef PCG_EXP_TAYLOR13
  double p = 1.6059043836821613e-10;             // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
#else
  double p = 2.511003840733968e-08;
  p = __builtin_fma(p, r, 2.7632640819274376e-07);
  p = __builtin_fma(p, r, 2.7557242367156263e-06);
  p = __builtin_fma(p, r, 2.4801487365644137e-05);
  p = __builtin_fma(p, r, 0.00019841269886564062);
  p = __builtin_fma(p, r, 0.001388888894778586);
  p = __builtin_fma(p, r, 0.008333333333322215);
  p = __builtin_fma(p, r, 0.041666666666522106);
  p = __builtin_fma(p, r, 0.16666666666666674);
  p = __builtin_fma(p, r, 0.500000000000001);
#endif
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}
// log(x) for finite, normal x > 0 (clamped below at DBL_MIN): x = m 2^e with m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716, odd Taylor series to s^21 (truncation 2e-17 relative),
// then e ln2 in two pieces.  ~1-2 ulp; ~40 VALU instructions against ~150 for the library log(), which carries
// the result in double-double and selects among zero / negative / inf / NaN / denormal inputs.
PCG_PK double log_pos(double x) {
  x = __builtin_fmax(x, 2.2250738585072014e-308);
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double s = div_fast(m - 1.0, m + 1.0);
  const double z = s * s;
  double p = 2.0 / 21.0;
  p = __builtin_fma(p, z, 2.0 / 19.0);
  p = __builtin_fma(p, z, 2.0 / 17.0);
  p = __builtin_fma(p, z, 2.0 / 15.0);
  p = __builtin_fma(p, z, 2.0 / 13.0);
  p = __builtin_fma(p, z, 2.0 / 11.0);
  p = __builtin_fma(p, z, 2.0 / 9.0);
  p = __builtin_fma(p, z, 2.0 / 7.0);
  p = __builtin_fma(p, z, 2.0 / 5.0);
  p = __builtin_fma(p, z, 2.0 / 3.0);
  const double lm = __builtin_fma(s * z, p, s + s);
  const double de = (double)e;
  return __builtin_fma(de, 6.93147180369123816490e-01, __builtin_fma(de, 1.90821492927058770002e-10, lm));
}
template <int W>
PCG_PK Pack<W> log_pos(const Pack<W>& a) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = log_pos(a.v[i]);
  return r;
}
// sin(pi x), cos(pi x) for x in [0, 2) (the Box-Muller angle 2 u, u in [0,1)): exact reduction to r in [-1/4, 1/4]
// by quarter turns, Taylor polynomials of sin / cos in y = pi r (|y| <= pi/4: truncation 5e-17 / 2e-18), quadrant
// fix-up.  ~35 VALU instructions against ~75 for the library sincospi (which also handles huge / non-finite x).
PCG_PK void sincospi_unit(double x, double& s, double& c) {
  const double n = __builtin_rint(x + x);  // quarter-turn index 0..4
  const double y = __builtin_fma(n, -0.5, x) * 3.14159265358979323846;
  const double z = y * y;
  double ps = -1.0 / 1307674368000.0;             // -1/15!
  ps = __builtin_fma(ps, z, 1.0 / 6227020800.0);  //  1/13!
  ps = __builtin_fma(ps, z, -1.0 / 39916800.0);   // -1/11!
  ps = __builtin_fma(ps, z, 1.0 / 362880.0);      //  1/9!
  ps = __builtin_fma(ps, z, -1.0 / 5040.0);       // -1/7!
  ps = __builtin_fma(ps, z, 1.0 / 120.0);         //  1/5!
  ps = __builtin_fma(ps, z, -1.0 / 6.0);          // -1/3!
  const double sy = __builtin_fma(y * z, ps, y);
  double pc = 1.0 / 20922789888000.0;             //  1/16!
  pc = __builtin_fma(pc, z, -1.0 / 87178291200.0);  // -1/14!
  pc = __builtin_fma(pc, z, 1.0 / 479001600.0);   //  1/12!
  pc = __builtin_fma(pc, z, -1.0 / 3628800.0);    // -1/10!
  pc = __builtin_fma(pc, z, 1.0 / 40320.0);       //  1/8!
  pc = __builtin_fma(pc, z, -1.0 / 720.0);        // -1/6!
  pc = __builtin_fma(pc, z, 1.0 / 24.0);          //  1/4!
  pc = __builtin_fma(pc, z, -0.5);
  const double cy = __builtin_fma(z, pc, 1.0);
  const int q = (int)n & 3;
  const double ss = (q & 1) ? cy : sy, cc = (q & 1) ? sy : cy;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// sqrt(x) for x >= 0 in the normal range (exact 0 -> 0; negative -> NaN like the library): hardware reciprocal-
// square-root estimate y (relative error <= 2^-24), g = x y, then ONE Newton step of the square root with the unrefined
// half-estimate, g + (y/2)(x - g^2): with y = (1 + e) / sqrt(x) the result is sqrt(x) (1 - 3 e^2 / 2 - e^3 / 2), i.e. a relative
// error of 1.5 e^2 <= 1.5 x 2^-48 = 5.3e-15 (~24 ulp; rounds 2-3 were within 1 ulp) for the measured |e| <= 2^-24 of v_rsq_f64 /
// v_rcp_f64 on gfx950 (tools/issuebench.hip over 4M arguments: 2^-24.4 -- a measurement of this ASIC, not an architectural
// guarantee: tests/test_gpu_parity.py::test_hardware_estimates_are_as_accurate_as_the_fast_math_assumes pins it).  RHS parity
// with the host sqrt() at 1e-12 holds with two decades to spare; last-bit agreement with it is given up.  6 VALU instructions -- one of them the 7.3-ns estimate
// (profiles/r4/issuebench.txt) -- against ~25 for the library sqrt(), whose extra work is the 2^+-256 rescaling for huge /
// denormal arguments.  Rounds 2-3 spent 9: a coupled Goldschmidt refinement of BOTH g and y/2 before the same correction,
// for a last-bit result nothing downstream of a 1e-7-accurate integrator needs; four_tank spends half of its instructions
// in this function.  The estimate is taken at x + 2^-1000, which is x itself for every x >= 2^-947 and turns the 0 * inf of
// an exact zero into 0 * 2^500 = 0 without a compare-and-select.
PCG_PK double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x + 0x1p-1000);
  const double g = x * y, h = 0.5 * y;
  return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
template <int W>
PCG_PK Pack<W> sqrt_pos(const Pack<W>& a) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = sqrt_pos(a.v[i]);
  return r;
}
template <int W>
PCG_PK Pack<W> rcp_fast(const Pack<W>& a) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = rcp_fast(a.v[i]);
  return r;
}
template <int W>
PCG_PK Pack<W> div_fast(const Pack<W>& a, const Pack<W>& b) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = div_fast(a.v[i], b.v[i]);
  return r;
}
template <int W>
PCG_PK Pack<W> div_fast(double a, const Pack<W>& b) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = div_fast(a, b.v[i]);
  return r;
}
template <int W>
PCG_PK Pack<W> exp_bounded(const Pack<W>& a) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = exp_bounded(a.v[i]);
  return r;
}

template <int W>
PCG_PK Pack<W> pow(const Pack<W>& a, double e) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = ::pow(a.v[i], e);
  return r;
}

// scalar overloads so model code can be written once for T = double and T = Pack<W>
PCG_PK double pk_get(double a, int) { return a; }
template <int W>
PCG_PK double pk_get(const Pack<W>& a, int i) {
  return a.v[i];
}

}  // namespace pcg
