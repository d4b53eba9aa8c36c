// pcg_seulex.hpp -- SEULEX-8: the integrator of the HEAVY envs of a PCG_INT_RODAS4 plan (cfg.coop_thr > 0).
//
// Why.  A launch of the adaptive Rosenbrock pair is as long as its heaviest env: on BASELINE configs[2] the mean env step
// takes 19 attempts, one env in a hundred takes 64 and the heaviest 105 -- 105 x 6 dependent (right-hand side + solve)
// stages on ONE lane, 305 us of a 320 us launch, whatever the other lanes do.  The reference's CVODES
// (integrator.py:163-182) integrates a stiff column at a cost that does not depend on its batch-mates; here the heavy env's
// chain has to get shorter.  Extrapolation is parallel by construction: extrapolated linearly implicit Euler with a fixed
// column of eight (Deuflhard's SEULEX without order selection) --
//   big step H from x:   row j = 0..7 with n_j = j + 1 sub-steps:  theta_j = n_j (1 / H),  W_j = theta_j I - J(x)  (J frozen),
//                        y = x;  n_j times:  d = W_j^-1 f(y),  y = y + d                                       -> T_j
//   Aitken-Neville in h: column c = 1..7, rows j >= c:   T_j <- fma(T_j - T_{j-1}, (n_j - c) / c, T_j)   (T_{j-1} of column c - 1)
//   new state T_7 (order 8), error estimate T_7 - T_7' (T_7' = the last row before the last column, order 7)
// -- gives row j to LANE j of eight (seulex8_try_lanes: the env's chain sees the deepest lane only, 8 sub-steps per big step,
// each lane factoring its own structured W_j with the model's ros_factor / ros_solve; the tableau by cross-lane reads),
// and crosses the heaviest env steps in <= 13 big steps instead of 105 attempts at the same accuracy class
// (tools/prototypes/seulex8_calib.py, profiles/r5/seulex8_calib.txt).  One lane running all eight rows one after the other
// (seulex8_try_serial: the classic one-env-per-lane kernels, pcg_integrate, the fused rollout) produces the SAME BITS: every
// entry of the tableau is one exactly specified IEEE operation on two entries of the previous column, whoever computes it.
//
// Which envs: a per-env rule, M::coop_key(kp, u, d1) >= cfg.coop_thr -- a fit of the Rosenbrock pair's attempts per env step
// in exact arithmetic (exponent extraction, no transcendental), so that the kernels and the oracle pick the same envs.
// Controller: mean-square norm with the end-point weights of rodas4(), tolerances sx::TOL x the plan's (the estimate belongs
// to the order-7 row), accept iff E2 < 1, factor Q(sx::SAFETY E2^(-1/16)) in [sx::FACMIN, sx::FACMAX] (<= 1 after a rejected
// big step), first step min(Q(sx::H0 x 5 h0), dt) with Hairer's h0; failure semantics as rodas4().
// Twin: seulex8() in oracle/pcg_oracle.c, operation for operation.
#pragma once

namespace pcg {

namespace sx {
constexpr int K = 8;           // rows of the tableau = lanes per env in the cooperative form
constexpr double TOL = 4.0;    // tolerance relative to the plan's (Rodas4's) rtol / atol
constexpr double H0 = 8.0;     // first big step relative to rodas4's first step
constexpr double SAFETY = 0.8, FACMAX = 4.0, FACMIN = 0.1;
}  // namespace sx

// model hook: M::coop_key(kp, u, d1) -- predicted attempts of the Rosenbrock pair for this env step
template <class M, class = void>
struct has_coop : tt::false_type {};
template <class M>
struct has_coop<M, tt::void_t<decltype(M::COOP)>> : tt::true_type {};

// the controller's verdict on one big step (twin of rodas4_factor)
PCG_DEV double seulex_factor(double E2, bool ok, bool rejected_last) {
  double fac = (E2 == E2) ? fmax(sx::FACMIN, ctrl_pow_e(E2, sx::SAFETY, 0.0625f, 0.0625)) : sx::FACMIN;  // NaN -> hardest shrink
  const double cap = ok ? (rejected_last ? 1.0 : sx::FACMAX) : 1.0;
  return fmin(cap, fac);
}

// first big step from Hairer's h0 (the statements of rodas4_h_init, another multiple)
PCG_DEV double seulex_h_init(double d0, double d1, double dt) {
#pragma clang fp contract(off)
  const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  return fmin(qtrunc6(sx::H0 * (5.0 * h0)), dt);
}

// Aitken-Neville weight of row j (0-based), column c: (n_j - c) / c, an IEEE division (the oracle's twin divides too)
PCG_DEV double seulex_w(int j, int c) { return (double)(j + 1 - c) / (double)c; }

// ---- one lane, all eight rows: the tableau is built row by row in place (R[c] = T_{j, c+1} of the row just finished) ----
template <class M, class K, class F>
PCG_DEV bool seulex8_try_serial(const K& kp, const typename M::Hold& hold, const F& f, const double (&x)[M::NX], double H,
                                double (&xn)[M::NX], double (&err)[M::NX]) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX;
  const double ih = rcp_ieee(H);  // (H is a positive normal number: == 1.0 / H)
  double R[sx::K][NX];
  bool lu_ok = true;
#pragma unroll
  for (int j = 0; j < sx::K; ++j) {
    typename M::RosFac Fj;
    M::ros_factor(kp, hold, x, (double)(j + 1) * ih, Fj);
    lu_ok = lu_ok && Fj.ok;
    double y[NX], d[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] = x[i];
#pragma unroll 1
    for (int s = 0; s <= j; ++s) {
      f(y, d);
      M::ros_solve(Fj, d);
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] = y[i] + d[i];
    }
    // row j of the tableau from row j - 1, in place: `old` walks the previous row's entries one column behind
    double old[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      old[i] = R[0][i];
      R[0][i] = y[i];
    }
#pragma unroll
    for (int c = 1; c <= j; ++c) {
      const double w = seulex_w(j, c);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const double keep = (c < j) ? R[c][i] : 0.0;  // T_{j-1, c+1} (row j - 1 has no column j + 1: nothing to keep)
        R[c][i] = __builtin_fma(R[c - 1][i] - old[i], w, R[c - 1][i]);
        old[i] = keep;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xn[i] = R[sx::K - 1][i];
    err[i] = R[sx::K - 1][i] - R[sx::K - 2][i];
  }
  return lu_ok;
}

// ---- eight lanes, one row each ----
// the value of the lane one below (row_shr:1 inside a row of 16 lanes: two 8-lane groups per row; lane 0 of a group reads
// its neighbour group's top lane or keeps its own value -- row 0 never uses what it reads)
PCG_DEV double sx_from_below(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// x, H and the result are replicated over the eight lanes of the env's group; j = lane & 7.  `xn` and `err` come back valid on
// EVERY lane (broadcast from the top lane), the return value is the conjunction of the eight factorisations.
template <class M, class K, class F>
PCG_DEV bool seulex8_try_lanes(const K& kp, const typename M::Hold& hold, const F& f, const double (&x)[M::NX], double H, int j,
                               const double (&w)[sx::K - 1], double (&xn)[M::NX], double (&err)[M::NX]) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX;
  const int lane = threadIdx.x & 63, top = lane | 7;
  const double ih = rcp_ieee(H);
  typename M::RosFac Fj;
  M::ros_factor(kp, hold, x, (double)(j + 1) * ih, Fj);
  double y[NX], d[NX], prev[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = x[i];
#pragma unroll 1
  for (int s = 0; s < sx::K; ++s) {  // the deepest lane sets the trip count of the wave
    if (s <= j) {
      f(y, d);
      M::ros_solve(Fj, d);
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] = y[i] + d[i];
    }
  }
#pragma unroll
  for (int c = 1; c < sx::K; ++c) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const double up = sx_from_below(y[i]);
      if (c == sx::K - 1) prev[i] = y[i];
      const double t = __builtin_fma(y[i] - up, w[c - 1], y[i]);
      y[i] = (j >= c) ? t : y[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xn[i] = __shfl(y[i], top);
    err[i] = __shfl(y[i] - prev[i], top);
  }
  const unsigned long long okm = __ballot(Fj.ok);
  return ((okm >> (lane & ~7)) & 0xFFull) == 0xFFull;
}

// ---- the loop (one lane per env), twin of rodas4() ----
template <class M, class K, class F, class EP>
PCG_DEV int seulex8(const K& kp, const typename M::Hold& hold, const F& f, const EP& ep, double (&x)[M::NX], int n, double dt,
                    double rtol, double atol, int max_steps, double h_init, int& nacc, int& nrej) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX;
  const double rt = sx::TOL * rtol, at = sx::TOL * atol;
  double xn[NX], err[NX];
  int acc = 0, rej = 0, status = 0;
  double H = h_init, t = 0.0;
  bool rejected_last = false;
  for (;;) {
    bool last = false;
    if (acc + rej >= max_steps) {
      status = 1;
      break;
    }
    if (t + H >= dt * (1.0 - 1e-14)) {
      H = dt - t;
      last = true;
    }
    const bool lu_ok = seulex8_try_serial<M>(kp, hold, f, x, H, xn, err);
    double E2 = ms_scaled_ep<NX>(ep, dt - (t + H), err, x, xn, n, rt, at);
    if (!lu_ok) E2 = __builtin_nan("");
    const bool ok = E2 < 1.0;
    const double fac = seulex_factor(E2, ok, rejected_last);
    if (ok) {
      t += H;
      H *= fac;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = xn[i];
      rejected_last = false;
      ++acc;
      if (last) break;
    } else {
      H *= fac;
      rejected_last = true;
      ++rej;
      if (!(H > 1e-13 * dt)) {
        status = 2;
        break;
      }
    }
  }
  nacc = acc;
  nrej = rej;
  return status;
}

// The per-env rule on one lane: true (and the env step integrated by SEULEX-8, status in `status`) when the model's
// cooperative rule picks this env, false when the pair is to integrate it.  The classic kernels, pcg_integrate and the
// fused rollout go through this function; the work-queue kernel evaluates the same rule when it parks an env.
template <class M, class K, class F, class EP>
PCG_DEV bool seulex8_if_heavy(const K& kp, const typename M::Hold& hold, const double (&u)[M::NA + M::NDM], const F& f,
                              const EP& ep, double (&x)[M::NX], int n, double dt, double rtol, double atol, int max_steps,
                              double coop_thr, int& nacc, int& nrej, int& status) {
  if constexpr (has_coop<M>::value) {
    if (coop_thr > 0.0) {
      constexpr int NX = M::NX;
      double f0[NX];
      f(x, f0);
      const double d0 = rms_scaled<NX>(x, x, x, n, rtol, atol);
      const double d1 = rms_scaled<NX>(f0, x, x, n, rtol, atol);
      if (M::coop_key(kp, u, d1) >= coop_thr) {
        status = seulex8<M>(kp, hold, f, ep, x, n, dt, rtol, atol, max_steps, seulex_h_init(d0, d1, dt), nacc, nrej);
        return true;
      }
    }
  }
  return false;
}

// ---- resumable form for the cooperative phase of the work-queue kernel: the state one GROUP of eight lanes carries ----
template <int NX>
struct SxGroup {
  double x[NX];
  double t, H;
  int acc, rej;
  bool rejected_last;
};
// one big step of the group's env.  Returns -1 to continue, else the final PCG_ST_* status (group-uniform).
template <class M, class K, class F, class EP>
PCG_DEV int seulex8_attempt_lanes(const K& kp, const typename M::Hold& hold, const F& f, const EP& ep, SxGroup<M::NX>& G, int j,
                                  const double (&w)[sx::K - 1], int n, double dt, double dt_edge, double h_floor, double rtol,
                                  double atol, int max_steps) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX;
  if (G.acc + G.rej >= max_steps) return PCG_ST_MAX_STEPS;
  bool last = false;
  double H = G.H;
  if (G.t + H >= dt_edge) {
    H = dt - G.t;
    last = true;
  }
  double xn[NX], err[NX];
  const bool lu_ok = seulex8_try_lanes<M>(kp, hold, f, G.x, H, j, w, xn, err);
  double E2 = ms_scaled_ep<NX>(ep, dt - (G.t + H), err, G.x, xn, n, sx::TOL * rtol, sx::TOL * atol);
  if (!lu_ok) E2 = __builtin_nan("");
  const bool ok = E2 < 1.0;
  const double fac = seulex_factor(E2, ok, G.rejected_last);
  G.t = ok ? G.t + H : G.t;
  G.H = H * fac;
#pragma unroll
  for (int i = 0; i < NX; ++i) G.x[i] = ok ? xn[i] : G.x[i];
  G.rejected_last = !ok;
  G.acc += ok ? 1 : 0;
  G.rej += ok ? 0 : 1;
  if (ok) return last ? PCG_ST_OK : -1;
  return !(G.H > h_floor) ? PCG_ST_UNDERFLOW : -1;
}

}  // namespace pcg
