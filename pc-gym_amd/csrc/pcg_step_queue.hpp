// pcg_step_queue.hpp -- adaptive (DOPRI5) env step with an in-workgroup work queue: lanes that finish their env
// early pull the next one instead of idling until the slowest lane of their wave is done.
//
// Why.  With one env per lane for a whole step, a wave's time is the MAXIMUM over its 64 lanes of the number of
// attempted RK steps; on BASELINE configs[2] (multistage extraction over the full action box: 23..132 attempts per
// env step, mean 72) the mean/max ratio inside a wave is 0.64 -- a third of the fp64 issue slots do nothing
// (profiles/r1/configs.jsonl).  Re-ordering the batch in HBM by cost was built in round 1 and lost to its scattered
// 8-byte accesses.  Here the re-balancing happens in LDS, and every HBM access stays coalesced:
//
//   phase 1  a 256-thread workgroup loads a tile of T envs (T <= 1024: four per lane) the usual SoA way, runs the
//            pre-integration half of the step (env_pre: action map, disturbances, t == 0 check) and parks the held
//            input in LDS [component][slot], together with the initial step size and a cost key.  The STATE stays where
//            it is, in HBM/L2: a slot costs 8 (NU + 1) + 12 bytes of LDS instead of 8 (NX + NU + 1) + 12, which is what
//            lets a tile hold four envs per lane (round 2's first build parked the state too and stopped at two:
//            BASELINE configs[4]'s extraction segment, 683 envs per workgroup, fell into two half-filled tiles and
//            lost the re-balancing gain);
//   sort     the tile's slots are ordered by DECREASING cost key (bitonic sort of 512 / 1024 packed 32-bit words in LDS):
//            longest-processing-time-first is what keeps the tail short when every lane only sees ~2 envs -- with FIFO
//            order the queue runs dry early and each wave still waits for its slowest last env (simulated on the
//            measured step counts: FIFO 1.12x, LPT by this key 1.36x, exact LPT 1.46x);
//   phase 2  every lane integrates one env at a time from the queue: one attempted RK step per loop iteration for
//            all busy lanes of the wave; idle lanes pop the next sorted slot (one wave-aggregated LDS atomic per
//            refill; refills are batched to >= 8 idle lanes because the refill code -- the state's NX 8-byte loads
//            from the tile's window of the batch (just read by phase 1: L2 hits) + one RHS evaluation -- runs for the
//            whole wave); the integrated state is written back to its env's place in the batch (NX 8-byte stores per
//            env step, against ~50,000 instructions of integration);
//   phase 3  the workgroup runs the post-integration half (env_post: SP slot, constraints, reward, noise,
//            observation) one env per lane again, reading the new state coalesced, and stores coalesced.
//
// Per-env arithmetic does not depend on which lane integrates an env or in which order: results are bitwise
// independent of the batch order (tested with a permuted batch).  The step-size controller, tolerances and failure
// semantics are those of dopri5() in pcg_integrators.hpp (same statements, kept in a resumable per-lane form).
// Reference: integrator.py:65-88 (adaptive explicit 5(4) pair, rtol = atol = 1e-8).
#pragma once

namespace pcg {

// Four waves share one tile.  A single-wave workgroup with its own 128-slot tile (no cross-wave barrier, queue
// head in a register) was built and measured as well: 1.05x over the classic kernel on BASELINE configs[2] against
// 1.13x for this shape -- the larger pool is worth more than the barriers cost (profiles/r2/queue_kernel.md).
constexpr int QBLOCK = 256;        // threads per workgroup (one wave per SIMD)
constexpr int QSORT = 2048;        // maximum tile = maximum sort width: eight envs per lane (the explicit pair stops at four)
constexpr int QSLOT_BITS = 11;     // slot index bits below the cost key in a sort word
constexpr int QREFILL = 8;         // idle lanes that trigger a refill in a well-filled tile (or: no busy lane left)

// model hook: a cheap, monotone proxy of the number of RK steps an env step will take (the sort key)
template <class M, class = void>
struct has_cost_key : tt::false_type {};
template <class M>
struct has_cost_key<M, tt::void_t<decltype(M::COST_KEY)>> : tt::true_type {};

// ---- what phase 2 knows about its tile (LDS arrays of the layout below) ------------------------------------------------
// LEAN layout (round 5; the host picks it where it is what lets the tile's STATE live in LDS too, QLayout::bytes_x): no
// first-step array (the step size is recomputed where an env is picked up: the same statements on the same values as where it
// was parked), the two step counts packed into one word (17 + 15 bits, saturating: exact up to max_steps = 131,071 -- the
// default budget is 100,000; writing them straight to the nsteps buffer from phase 2 was measured first: 4-byte scattered
// stores cost a 64-byte read-modify-write each, 90 MB per launch of configs[4]'s segment), and of the held input only the
// action and the CONFIGURED disturbance values (the model's other disturbance inputs are plan constants).
template <class M>
struct QTile {
  const double* us;       // [nus][T]
  const double* hs;       // [T] first step size, or null (lean)
  int32_t *accs, *rejs;   // [T]; lean: accs holds both counts packed, rejs is null
  uint8_t* flag;          // [T] bits 0-1: PCG_ST_* of the integration, bit 2: pre-step "done"
  int T;
  bool lean;
  int nd;                          // configured disturbances (lean: rows NA .. NA + nd - 1 of us)
  const PCG_CONSTANT int32_t* d_slot;    // which model input each of them drives
  const PCG_CONSTANT double* d_default;  // the model's disturbance inputs when not configured
  static constexpr int ACC_BITS = 17, ACC_MAX = (1 << 17) - 1, REJ_MAX = (1 << 15) - 1;
  // the held input of a slot
  PCG_DEV void load_u(int slot, double (&u)[M::NA + M::NDM]) const {
    constexpr int NA = M::NA, NDM = M::NDM, NU = NA + NDM;
    if (!lean) {
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = us[(size_t)i * T + slot];
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) u[i] = us[(size_t)i * T + slot];
#pragma unroll
    for (int j = 0; j < NDM; ++j) u[NA + j] = d_default[j];
#pragma unroll
    for (int k = 0; k < NDM; ++k)
      if (k < nd) {
        const double v = us[(size_t)(NA + k) * T + slot];
        const int sl = d_slot[k];
#pragma unroll
        for (int j = 0; j < NDM; ++j) u[NA + j] = (j == sl) ? v : u[NA + j];
      }
  }
  // a finished env's result bookkeeping
  PCG_DEV void finish(int slot, int acc, int rej, int st) const {
    if (lean) {
      accs[slot] = (acc < ACC_MAX ? acc : ACC_MAX) | ((rej < REJ_MAX ? rej : REJ_MAX) << ACC_BITS);
    } else {
      accs[slot] = acc;
      rejs[slot] = rej;
    }
    flag[slot] |= (uint8_t)st;
  }
  PCG_DEV void counts(int slot, int& acc, int& rej) const {
    const int v = accs[slot];
    acc = lean ? (v & ACC_MAX) : v;
    rej = lean ? (int)((unsigned)v >> ACC_BITS) : rejs[slot];
  }
};

// ---- DOPRI5 in resumable form: the state one lane carries for the env it is integrating -------------------------
template <int NX>
struct DpLane {
  double x[NX], k1[NX];
  double t, h;
  int acc, rej;
  bool rejected_last;
};

// k1 = f(x) and the initial step size (Hairer, Norsett & Wanner II.4) -- the statements of dopri5() before its loop
template <int NX, class F>
PCG_DEV double dopri5_h_init(const F& f, const double (&x)[NX], double (&k1)[NX], int n, double dt, double rtol,
                             double atol, double& d1_out) {
#pragma clang fp contract(off)
  double y[NX], w[NX];
  f(x, k1);
  const double d0 = rms_scaled<NX>(x, x, x, n, rtol, atol);
  const double d1 = rms_scaled<NX>(k1, x, x, n, rtol, atol);
  d1_out = d1;
  double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  h0 = fmin(h0, dt);
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = axpy(h0, k1[i], x[i]);
  f(y, w);
#pragma unroll
  for (int i = 0; i < NX; ++i) w[i] -= k1[i];
  const double d2 = rms_scaled<NX>(w, x, x, n, rtol, atol) / h0;
  const double dm = fmax(d1, d2);
  const double h1 = (dm <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : ctrl_pow(dm * dm * 1e4, 1.0);
  return fmin(qtrunc6(fmin(100.0 * h0, h1)), dt);
}

// ... with k1 = f(x) already there (the barrier-free rollout has it from the guard's evaluation at the start state: the same
// bits): the statements of dopri5_h_init() after the first right-hand side
template <int NX, class F>
PCG_DEV double dopri5_h_init_k1(const F& f, const double (&x)[NX], const double (&k1)[NX], int n, double dt, double rtol,
                                double atol) {
#pragma clang fp contract(off)
  double y[NX], w[NX];
  const double d0 = rms_scaled<NX>(x, x, x, n, rtol, atol);
  const double d1 = rms_scaled<NX>(k1, x, x, n, rtol, atol);
  double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  h0 = fmin(h0, dt);
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = axpy(h0, k1[i], x[i]);
  f(y, w);
#pragma unroll
  for (int i = 0; i < NX; ++i) w[i] -= k1[i];
  const double d2 = rms_scaled<NX>(w, x, x, n, rtol, atol) / h0;
  const double dm = fmax(d1, d2);
  const double h1 = (dm <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : ctrl_pow(dm * dm * 1e4, 1.0);
  return fmin(qtrunc6(fmin(100.0 * h0, h1)), dt);
}

// one attempted step (the body of dopri5()'s loop).  Returns -1 to continue, else the final PCG_ST_* status
// (PCG_ST_OK: reached dt; on failure the caller poisons the state).
template <int NX, class F>
PCG_DEV int dopri5_attempt(const F& f, DpLane<NX>& L, int n, double dt, double dt_edge, double h_floor, double rtol,
                           double atol, int max_steps) {
#pragma clang fp contract(off)
  if (L.acc + L.rej >= max_steps) return PCG_ST_MAX_STEPS;
  bool last = false;
  double h = L.h;
  if (L.t + h >= dt_edge) {  // dt (1 - 1e-14), folded on the host: no scalar fp64 unit -> it would sit in a VGPR
    h = dt - L.t;
    last = true;
  }
  using namespace dp5;
  // ACCUMULATOR FORM.  The stage sums are left-to-right fused multiply-add chains (dp5_row(): for NX > 4 starting at x with the step
  // size folded into the coefficients): once k4 is there, the partial sums of the rows still to come (stage 6, the 5th-order
  // solution, the error estimate) are formed and k2..k4 are dead; k5 and k6 are folded in as they arrive.  The same
  // operations on the same operands in the same order as dopri5()'s rows -- bit for bit -- with at most six NX-vectors
  // alive instead of eight: the 20-state cascade's attempt carried 216 8-byte moves to and from the accumulation
  // registers (256 + 166 registers: one wave per SIMD) for 1390 fp64 operations; the 10-state one fits either way.
  constexpr bool FOLD = dp5_fold(NX);
  double y[NX], kk[NX], k2[NX], k3[NX], s6[NX], s7[NX], se[NX];
  const double (&x)[NX] = L.x;
  const double (&k1)[NX] = L.k1;
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = dp5_row<FOLD>(x[i], h, a21, k1[i]);
  f(y, k2);
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = dp5_row<FOLD>(x[i], h, a31, k1[i], a32, k2[i]);
  f(y, k3);
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = dp5_row<FOLD>(x[i], h, a41, k1[i], a42, k2[i], a43, k3[i]);
  f(y, kk);  // k4
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    y[i] = dp5_row<FOLD>(x[i], h, a51, k1[i], a52, k2[i], a53, k3[i], a54, kk[i]);
    if constexpr (FOLD) {  // rows 6 and 7 up to k4, starting at x
      s6[i] = xlc4(x[i], h * a61, k1[i], h * a62, k2[i], h * a63, k3[i], h * a64, kk[i]);
      s7[i] = xlc3(x[i], h * b1, k1[i], h * b3, k3[i], h * b4, kk[i]);
      se[i] = lc3(h * e1, k1[i], h * e3, k3[i], h * e4, kk[i]);
    } else {
      s6[i] = lc4(a61, k1[i], a62, k2[i], a63, k3[i], a64, kk[i]);
      s7[i] = lc3(b1, k1[i], b3, k3[i], b4, kk[i]);
      se[i] = lc3(e1, k1[i], e3, k3[i], e4, kk[i]);
    }
  }
  f(y, k2);  // k5
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    if constexpr (FOLD) {
      y[i] = __builtin_fma(h * a65, k2[i], s6[i]);
      s7[i] = __builtin_fma(h * b5, k2[i], s7[i]);
      se[i] = __builtin_fma(h * e5, k2[i], se[i]);
    } else {
      y[i] = axpy(h, __builtin_fma(a65, k2[i], s6[i]), x[i]);
      s7[i] = __builtin_fma(b5, k2[i], s7[i]);
      se[i] = __builtin_fma(e5, k2[i], se[i]);
    }
  }
  f(y, k3);  // k6
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    if constexpr (FOLD) {
      y[i] = __builtin_fma(h * b6, k3[i], s7[i]);
      se[i] = __builtin_fma(h * e6, k3[i], se[i]);
    } else {
      y[i] = axpy(h, __builtin_fma(b6, k3[i], s7[i]), x[i]);
      se[i] = __builtin_fma(e6, k3[i], se[i]);
    }
  }
  f(y, kk);  // k7 at the 5th-order solution (FSAL)
  // error estimate and its scaled mean square in one pass (ms_scaled() without the intermediate vector)
  double E2 = 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const double wi = FOLD ? __builtin_fma(h * e7, kk[i], se[i]) : h * __builtin_fma(e7, kk[i], se[i]);
    const double sc = atol + rtol * fmax(fabs(x[i]), fabs(y[i]));
    const double r = wi * fast_rcp(sc);  // sc > 0
    E2 += (i < n) ? r * r : 0.0;
  }
  E2 = E2 / n;  // accept iff E = sqrt(E2) < 1
  // One evaluation of the controller for both outcomes, the outcome applied by selects: in a wave of 64 lanes some
  // lane rejects in almost every iteration (~5 % of the attempts each), so an `if (accept) ... else ...` with the
  // controller in both arms executed both arms -- ~120 instructions per iteration (same values, bit for bit).
  const bool ok = E2 < 1.0;
  double fac = (E2 == E2) ? fmax(0.2, ctrl_pow(E2, 0.9)) : 0.2;  // NaN -> hardest shrink
  const double cap = ok ? ((L.rejected_last) ? 1.0 : 10.0) : 1.0;
  fac = fmin(cap, fac);
  L.t = ok ? L.t + h : L.t;
  L.h = h * fac;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    L.x[i] = ok ? y[i] : L.x[i];
    L.k1[i] = ok ? kk[i] : L.k1[i];
  }
  L.rejected_last = !ok;
  L.acc += ok ? 1 : 0;
  L.rej += ok ? 0 : 1;
  if (ok) return last ? PCG_ST_OK : -1;
  return !(L.h > h_floor) ? PCG_ST_UNDERFLOW : -1;  // 1e-13 dt: step-size underflow (NaN state / blow-up)
}

// ---- Rodas4 in resumable form (models with structured W: pcg_integrators.hpp).  Between attempts a lane only carries
// the state: f(x) is re-evaluated at the start of an attempt (the same bits as carrying it, one RHS evaluation more on
// the rare rejected attempt, ten doubles fewer alive across the six stages).
template <int NX>
struct R4Lane {
  double x[NX];
  double t, h;
  int acc, rej;
  bool rejected_last;
};
template <int NX, int INTEG>
struct QLaneSel { using type = DpLane<NX>; };
template <int NX>
struct QLaneSel<NX, PCG_INT_RODAS4> { using type = R4Lane<NX>; };
template <int NX>
struct QLaneSel<NX, PCG_INT_RODAS5> { using type = R4Lane<NX>; };

// one attempted step: the body of ros_pair()'s loop (INTEG: PCG_INT_RODAS4 or PCG_INT_RODAS5).  Returns -1 to continue, else
// the final PCG_ST_* status.
template <class M, int INTEG, class K, class F, class EP>
PCG_DEV int rodas4_attempt(const K& kp, const typename M::Hold& hold, const F& f, const EP& ep, R4Lane<M::NX>& L, int n,
                           double dt, double dt_edge, double h_floor, double rtol, double atol, int max_steps) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX;
  if (L.acc + L.rej >= max_steps) return PCG_ST_MAX_STEPS;
  bool last = false;
  double h = L.h;
  if (L.t + h >= dt_edge) {
    h = dt - L.t;
    last = true;
  }
  double f0[NX], xn[NX], err[NX];
  f(L.x, f0);
  const RosStructured<M, K> ls{kp, hold, {}};
  const bool lu_ok = ros_pair_try<INTEG, NX>(f, ls, L.x, f0, h, xn, err);
  double E2 = ms_scaled_ep<NX>(ep, dt - (L.t + h), err, L.x, xn, n, rtol, atol, ros_ep_cap<INTEG>(dt - (L.t + h), h));
  if (!lu_ok) E2 = __builtin_nan("");
  const bool ok = E2 < 1.0;
  const double fac = ros_pair_factor<INTEG>(E2, ok, L.rejected_last);
  L.t = ok ? L.t + h : L.t;
  L.h = h * fac;
#pragma unroll
  for (int i = 0; i < NX; ++i) L.x[i] = ok ? xn[i] : L.x[i];
  L.rejected_last = !ok;
  L.acc += ok ? 1 : 0;
  L.rej += ok ? 0 : 1;
  if (ok) return last ? PCG_ST_OK : -1;
  return !(L.h > h_floor) ? PCG_ST_UNDERFLOW : -1;
}

#ifndef PCG_Q_TAIL
#define PCG_Q_TAIL 2
#endif
constexpr bool q_tail(bool fixup) { return PCG_Q_TAIL >= 2 || (PCG_Q_TAIL == 1 && fixup); }

// ---- phase 2: the work queue of one tile.  (Tried as a real, non-inlined function so that the loop would own the
// whole register file: the call ABI's save / restore made it worse -- 772 B of scratch against 140.)
template <class M, int INTEG = PCG_INT_DOPRI5, int QB = QBLOCK, bool COMPACT = false>
PCG_DEV void queue_integrate(typename M::CKP* kpp, double* xg, int64_t xstride, const QTile<M>& Q,
                                                          const uint32_t* sortbuf,
                                                          int32_t* next, int n, int refill, double dt, double dt_edge,
                                                          double h_floor, double rtol, double atol, int max_steps,
                                                          double ep_c = 0.0, int ep_kmax = 0, int prio_h = 0, int rot = 0,
                                                          unsigned long long* qst = nullptr, const int32_t* eidx = nullptr, int first = 0) {
  const int T = Q.T;
  constexpr int NX = M::NX, NU = M::NA + M::NDM;
  // COMPACT (fix-up launch of a guarded plan): the slots are a compact list of marked envs, eidx[slot] is the env's position
  // in the state window xg; otherwise a slot is its own position
  auto xpos = [&](int sl) -> size_t {
    if constexpr (COMPACT) return (size_t)eidx[sl];
    else return (size_t)sl;
  };
#ifdef PCG_QSTATS  // measurement build (tools/queue_probe.py): per-wave counts of what the loop below did
  unsigned long long qs_iter = 0, qs_att = 0, qs_busy = 0, qs_refill = 0, qs_refill_clk = 0, qs_pop = 0;
#endif
  typename M::CKP& kp = *kpp;
  const int tid = threadIdx.x;
  typename QLaneSel<NX, INTEG>::type L;
  // sorted position of the lane's current env (the sort is by decreasing cost: small = heavy).  `rot` shifts which wave
  // starts with the heaviest 64: two workgroups that share a CU put them on different SIMDs (see q_prio)
  // first > 0: the first `first` sorted slots are the tile's heavy envs, integrated by the cooperative phase (coop_integrate);
  // its waves arrive here late and at different times, so nothing is handed out directly: every lane starts idle and pops
  int pos = first > 0 ? n : ((tid + rot) & (QB - 1));
  int slot = pos < n ? (int)(sortbuf[pos] & (QSORT - 1)) : -1;  // QSORT - 1 == the QSLOT_BITS mask
  bool fresh = slot >= 0;
  bool wave_hi = false;
  bool drained = first > 0 ? n <= first : n <= QB;  // wave-uniform: the queue has nothing (left) for this wave
  // every spin is bounded: a lane integrates at most two envs' worth of its tile share plus the refill rounds; the
  // bound is never reached by a correct run (max_steps bounds each env) and turns a logic error into a flagged
  // PCG_ST_MAX_STEPS result instead of a hung GPU
  const long long cap64 = ((long long)max_steps + 4) * ((T + QB - 1) / QB + 1) + 4 * T;
  const int iter_cap = cap64 > 0x7fffff00LL ? 0x7fffff00 : (int)cap64;
  for (int iter = 0;; ++iter) {
    if (iter > iter_cap) {
      if (slot >= 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xg[(size_t)i * xstride + xpos(slot)] = __builtin_nan("");
        Q.finish(slot, L.acc, L.rej, PCG_ST_MAX_STEPS);
      }
      break;
    }
#ifdef PCG_QSTATS
    ++qs_iter;
    const unsigned long long qs_any_fresh = __ballot(fresh);
    const unsigned long long qs_c0 = qs_any_fresh ? wall_clock64() : 0ull;
#endif
    if (fresh) {  // (re)fill: state from the batch (xg = &x[0][base of the tile]), held input from the slot, k1 = f(x)
#pragma unroll
      for (int i = 0; i < NX; ++i) L.x[i] = xg[(size_t)i * xstride + xpos(slot)];
      double u[NU];
      Q.load_u(slot, u);
      L.t = 0.0;
      L.acc = L.rej = 0;
      L.rejected_last = false;
      const typename M::Hold hold = M::hold(kp, u);
      const RhsFn<M> f{kp, hold};
      if constexpr (INTEG == PCG_INT_DOPRI5) {
        if (Q.lean) {  // (the statements of park(): k1 = f(x) and the first step size)
          double d1;
          L.h = dopri5_h_init<NX>(f, L.x, L.k1, NX, dt, rtol, atol, d1);
        } else {
          L.h = Q.hs[slot];
          f(L.x, L.k1);
        }
      } else {
        if (Q.lean) {
          double f0[NX], d1;
          f(L.x, f0);
          L.h = rodas4_h_init<NX, INTEG>(L.x, f0, NX, dt, rtol, atol, d1);
        } else {
          L.h = Q.hs[slot];
        }
      }
      fresh = false;
    }
#ifdef PCG_QSTATS
    if (qs_any_fresh) {
      ++qs_refill;
      qs_refill_clk += wall_clock64() - qs_c0;
    }
#endif
    const bool busy = slot >= 0;
    const unsigned long long bm = __ballot(busy);
    const int n_idle = 64 - __popcll(bm);
    // the shared queue head is only touched when this wave could use it (>= QREFILL idle lanes, or nothing left
    // in flight) and has not seen it empty yet: the steady-state iteration does no LDS access at all
    if (!drained && (n_idle >= refill || bm == 0ull)) {
#ifdef PCG_QSTATS
      ++qs_pop;
#endif
      // idle lanes pop: one LDS atomic per wave, lane r of the idle set takes sorted position head + r
      const unsigned long long im = ~bm;
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
      int got = 0;
      if (!busy && rank == 0) got = atomicAdd(next, n_idle);
      got = __shfl(got, __ffsll((long long)im) - 1);
      drained = got + n_idle >= n;  // the head only grows: once past n it stays there
      if (!busy) {
        const int j = got + rank;
        if (j < n) {
          slot = (int)(sortbuf[j] & (QSORT - 1));
          pos = j;
          fresh = true;
        }
      }
      if (got < n) continue;
    }
    if (bm == 0ull) break;  // nothing in flight in this wave and the queue is empty
    // ---- TAIL: the queue holds nothing more for this wave, every busy lane finishes the env it has ----
    // Once `drained` no lane of this wave will be handed another env: what is left is the classic one-env-per-lane loop, and
    // it is run as such -- the held input and the model's per-step constants in registers, no ballot, no queue head, no
    // priority bookkeeping per attempt.  The attempt is the same function on the same lane state: the same bits.  A wave
    // that has its SIMD to itself issues ONE instruction every four cycles whatever its kind, so the ~100 instructions of
    // bookkeeping per iteration cost it as much as 100 fused multiply-adds.  Measured (profiles/r5/queue_tail.txt): the
    // default cstr plan on the full x0 box 175 -> 167 us per step (its fix-up launch is one lane's chain across an ignition
    // front), me10 610 -> 581, me20 334 -> 320, me10_ros5 223.5 -> 217.5, configs[4]'s shard 332 -> 329.
    // PCG_Q_TAIL (A/B builds): 0 = off, 1 = the fix-up launch only, 2 = every work-queue kernel (the default).
    if constexpr (q_tail(COMPACT)) {
      if (drained) {  // (wave-uniform; no lane is `fresh` here: a fresh lane was refilled at the top of this iteration)
        if (prio_h > 0 && !wave_hi) __builtin_amdgcn_s_setprio(0);
        if (busy) {
          double u[NU];
          Q.load_u(slot, u);
          const typename M::Hold hold = M::hold(kp, u);
          const RhsFn<M> f{kp, hold};
          int st;
          if constexpr (is_ros_pair(INTEG)) {
            const EpWeights<M, typename M::CKP> ep{kp, u, ep_c, ep_kmax};
            do st = rodas4_attempt<M, INTEG>(kp, hold, f, ep, L, NX, dt, dt_edge, h_floor, rtol, atol, max_steps);
            while (st < 0);
          } else {
            do st = dopri5_attempt<NX>(f, L, NX, dt, dt_edge, h_floor, rtol, atol, max_steps);
            while (st < 0);
          }
          poison_if_failed<NX>(st, L.x);
#pragma unroll
          for (int i = 0; i < NX; ++i) xg[(size_t)i * xstride + xpos(slot)] = L.x[i];
          Q.finish(slot, L.acc, L.rej, st);
          slot = -1;
        }
        break;
      }
    }
#ifdef PCG_QSTATS
    ++qs_att;
    qs_busy += __popcll(bm);
#endif
    if (prio_h > 0) {  // a launch is as long as its heaviest env: the wave that carries one goes first on its SIMD
      const bool hi = __ballot(busy && pos < prio_h) != 0ull;
      if (hi != wave_hi) {
        if (hi) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        wave_hi = hi;
      }
    }
    if (busy) {
      // the held input stays in LDS between attempts (a few ds_read per ~1000-instruction attempt) instead of in
      // registers: the resumable loop sits right at the 256-register budget
      double u[NU];
      Q.load_u(slot, u);
      const typename M::Hold hold = M::hold(kp, u);
      const RhsFn<M> f{kp, hold};
      int st;
      if constexpr (is_ros_pair(INTEG)) {
        const EpWeights<M, typename M::CKP> ep{kp, u, ep_c, ep_kmax};
        st = rodas4_attempt<M, INTEG>(kp, hold, f, ep, L, NX, dt, dt_edge, h_floor, rtol, atol, max_steps);
      } else {
        st = dopri5_attempt<NX>(f, L, NX, dt, dt_edge, h_floor, rtol, atol, max_steps);
      }
      if (st >= 0) {  // finished (or gave up): park the result, free the lane
        poison_if_failed<NX>(st, L.x);
#pragma unroll
        for (int i = 0; i < NX; ++i) xg[(size_t)i * xstride + xpos(slot)] = L.x[i];
        Q.finish(slot, L.acc, L.rej, st);
        slot = -1;
      }
    }
  }
#ifdef PCG_QSTATS
  if ((tid & 63) == 0 && qst) {
    qst[6] = qs_iter, qst[7] = qs_att, qst[8] = qs_busy, qst[9] = qs_refill, qst[10] = qs_refill_clk, qst[11] = qs_pop;
  }
#endif
}

// ---- phase 2a: the cooperative phase of a Rodas4 tile (pcg_seulex.hpp).  The tile's heavy envs (the first nh sorted slots)
// are integrated by SEULEX-8 with EIGHT LANES PER ENV: a wave carries eight envs, one per group of eight lanes, one big step
// of every busy group per loop iteration; a group that finishes its env pops the next heavy one (one LDS atomic per wave and
// iteration with an idle group, heaviest first).  A wave leaves when the heavy queue is empty and its groups are done, and
// joins the pair's queue (queue_integrate).  The env's state and step size are replicated over its group's lanes.
template <class M, int QB, bool XL = false>
PCG_DEV void coop_integrate(typename M::CKP* kpp, double* xg, int64_t xstride, const QTile<M>& Q,
                            const uint32_t* sortbuf, int32_t* cnext, int nh,
                            double dt, double dt_edge, double h_floor, double rtol, double atol, int max_steps, double ep_c,
                            int ep_kmax, unsigned long long* qst = nullptr) {
  const int T = Q.T;
  (void)T;
  constexpr int NX = M::NX, NU = M::NA + M::NDM;
#ifdef PCG_QSTATS  // measurement build (tools/queue_probe.py): big steps this wave executed, busy groups summed over them
  unsigned long long qs_big = 0, qs_groups = 0;
#endif
  typename M::CKP& kp = *kpp;
  const int lane = threadIdx.x & 63, j = lane & 7;
  const unsigned long long below = (1ull << (lane & ~7)) - 1ull;  // the lanes of the groups before this one
  double w[sx::K - 1];
#pragma unroll
  for (int c = 1; c < sx::K; ++c) w[c - 1] = seulex_w(j, c);
  SxGroup<NX> G;
  int slot = -1;
  bool drained = false;  // wave-uniform
  // bounded like the pair's queue: max_steps bounds each env, a wave carries at most nh envs one after the other
  const long long cap64 = ((long long)max_steps + 4) * ((long long)nh + 1);
  const int iter_cap = cap64 > 0x7fffff00LL ? 0x7fffff00 : (int)cap64;
  for (int iter = 0;; ++iter) {
    if (iter > iter_cap) {
      if (slot >= 0 && j == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xg[(size_t)i * xstride + slot] = __builtin_nan("");
        Q.finish(slot, G.acc, G.rej, PCG_ST_MAX_STEPS);
      }
      break;
    }
    const unsigned long long idle = __ballot(slot < 0) & 0x0101010101010101ull;  // one bit per idle group
    if (!drained && idle != 0ull) {
      const int n_idle = __popcll(idle);
      int got = 0;
      if (lane == __ffsll((long long)idle) - 1) got = atomicAdd(cnext, n_idle);
      got = __shfl(got, __ffsll((long long)idle) - 1);
      drained = got + n_idle >= nh;
      if (slot < 0) {
        const int q = got + __popcll(idle & below);
        if (q < nh) {  // pick up: state from the tile (LDS or the batch), first big step from the slot
          slot = (int)(sortbuf[q] & (QSORT - 1));
#pragma unroll
          for (int i = 0; i < NX; ++i) G.x[i] = xg[(size_t)i * xstride + slot];
          if (Q.lean) {  // (the statements of park() for a heavy env)
            double u0[NU], f0[NX];
            Q.load_u(slot, u0);
            const typename M::Hold hold0 = M::hold(kp, u0);
            const RhsFn<M> ff{kp, hold0};
            ff(G.x, f0);
            const double d0 = rms_scaled<NX>(G.x, G.x, G.x, NX, rtol, atol);
            const double d1 = rms_scaled<NX>(f0, G.x, G.x, NX, rtol, atol);
            G.H = seulex_h_init(d0, d1, dt);
          } else {
            G.H = Q.hs[slot];
          }
          G.t = 0.0;
          G.acc = G.rej = 0;
          G.rejected_last = false;
        }
      }
    }
    if (__ballot(slot >= 0) == 0ull) {
      if (drained) break;
      continue;
    }
#ifdef PCG_QSTATS
    ++qs_big;
    qs_groups += __popcll(__ballot(slot >= 0) & 0x0101010101010101ull);
#endif
    if (slot >= 0) {
      double u[NU];
      Q.load_u(slot, u);
      const typename M::Hold hold = M::hold(kp, u);
      const RhsFn<M> f{kp, hold};
      const EpWeights<M, typename M::CKP> ep{kp, u, ep_c, ep_kmax};
      const int st = seulex8_attempt_lanes<M>(kp, hold, f, ep, G, j, w, NX, dt, dt_edge, h_floor, rtol, atol, max_steps);
      if (st >= 0) {  // finished (or gave up): lane 0 of the group parks the result, the group is free
        poison_if_failed<NX>(st, G.x);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xg[(size_t)i * xstride + slot] = G.x[i];
          Q.finish(slot, G.acc, G.rej, st);
        }
        slot = -1;
      }
    }
  }
#ifdef PCG_QSTATS
  if (lane == 0 && qst) qst[14] = qs_big, qst[15] = qs_groups;
#endif
}

// ---- the tile's sort: bitonic network over S = E * QB packed words (QB threads), DESCENDING, E words per thread in registers ----
// Element i = tid * E + r.  A compare-exchange at distance j pairs i with i ^ j:
//   j < E          both in the same thread's registers: no data movement
//   E <= j < 64 E  the partner sits (j / E) lanes away in the same wave: one cross-lane read, no barrier
//   j >= 64 E      another wave: through LDS, two barriers -- 3 of the 55 steps of a 1024-slot tile
// Rounds 2-3 ran all 55 steps through LDS with a barrier each: 16 us of the 20-state cascade's 340 us launch, 9 of the
// 10-state one's (tools/queue_probe.py); this form takes ~3.  Keys are unique (the slot index is in the low bits).
template <int E, int QB>
PCG_DEV void sort_tile(uint32_t* sortbuf) {
  constexpr int S = E * QB;
  const int tid = threadIdx.x;
  uint32_t v[E];
#pragma unroll
  for (int r = 0; r < E; ++r) v[r] = sortbuf[tid * E + r];
#pragma unroll
  for (int k = 2; k <= S; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64 * E) {  // cross-wave
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) sortbuf[tid * E + r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int i = tid * E + r;
          const uint32_t vp = sortbuf[i ^ j];
          const bool want_max = ((i & j) == 0) == ((i & k) == 0);
          v[r] = want_max ? (v[r] > vp ? v[r] : vp) : (v[r] < vp ? v[r] : vp);
        }
      } else if (j >= E) {  // cross-lane
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int i = tid * E + r;
          const uint32_t vp = (uint32_t)__shfl_xor((int)v[r], j / E);
          const bool want_max = ((i & j) == 0) == ((i & k) == 0);
          v[r] = want_max ? (v[r] > vp ? v[r] : vp) : (v[r] < vp ? v[r] : vp);
        }
      } else {  // in registers
#pragma unroll
        for (int r = 0; r < E; ++r) {
          if ((r & j) == 0) {
            const int i = tid * E + r;
            const uint32_t a = v[r], b = v[r | j];
            const bool desc = (i & k) == 0;
            const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
            v[r] = desc ? hi : lo;
            v[r | j] = desc ? lo : hi;
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; ++r) sortbuf[tid * E + r] = v[r];
  __syncthreads();
}

// LDS layout of one tile (T slots):
//   us[nus][T] | hs[T] (-) | sortbuf[sort_words(T)] u32 | acc[T] i32 | rej[T] i32 (-) | flag[T] u8 (padded to 8) | 4 counters | xs ...
// (-) not in the LEAN layout (QTile), where nus = NA + configured disturbances instead of NA + NDM and acc holds both counts.
template <class M>
struct QLayout {
  static constexpr int NU = M::NA + M::NDM;
  PCG_HD static int sort_words(int T) { return T <= QSORT / 4 ? QSORT / 4 : (T <= QSORT / 2 ? QSORT / 2 : QSORT); }
  PCG_HD static size_t bytes_l(int T, bool lean, int nus) {
    return sizeof(double) * (size_t)(lean ? nus : NU + 1) * T + sizeof(uint32_t) * (size_t)sort_words(T) +
           sizeof(int32_t) * (lean ? 1 : 2) * (size_t)((T + 1) & ~1) + (size_t)((T + 7) & ~7) + 16;
  }
  PCG_HD static size_t bytes(int T) { return bytes_l(T, false, NU); }
  // with the tile's STATE parked in LDS too (xs[NX][T], round 3): every access of the batch stays coalesced -- phase 1
  // reads x once, phase 3 writes it once -- instead of NX scattered 8-byte loads / stores per env whenever a lane picks
  // up / finishes one (PMC, round 2: 2.3-4.3x the algorithmic bytes).  Used when it fits beside the other workgroups.
  PCG_HD static size_t bytes_x(int T) { return bytes(T) + sizeof(double) * (size_t)M::NX * T; }
  PCG_HD static size_t bytes_x_lean(int T, int nus) { return bytes_l(T, true, nus) + sizeof(double) * (size_t)M::NX * T; }
};

// WAVES: waves per SIMD = workgroups per CU the register allocator is asked to leave room for (0 = the model's default,
// wpe()).  The Rosenbrock loop wants ~316 registers: at two waves per SIMD it spills 34 scratch accesses per attempt, at
// one (256 VGPRs + AGPRs) none -- the instantiation for launches that fit ONE tile per CU (me10 at B = 2^18: 0.361 -> 0.337 ms)
// QB: threads per workgroup.  256 = one wave per SIMD and workgroup, as many workgroups per CU as the registers allow;
// 512 = the two waves of every SIMD belong to ONE workgroup and share ONE tile of twice the size (see pcg_abi.hip:
// queue launch geometry): a pool twice as deep for the same lanes, and no wave left alone behind its SIMD-mate's tile.
// FIX: the fix-up launch of a guarded plan (its own instantiation: the launches of the adaptive plans carry none of it)
template <class M, bool PER_ENV_T, bool EXTRAS, int INTEG = PCG_INT_DOPRI5, int WAVES = 0, int QB = QBLOCK, bool FIX = false>
__global__ __launch_bounds__(QB, WAVES > 0 ? WAVES : wpe(M::NX, PCG_INT_DOPRI5, false)) void step_kernel_queue(const StepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  static_assert(!M::DYNAMIC, "the work-queue kernel is built for the fixed-size models");
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA, NDM = M::NDM, NU = NA + NDM;
  const int T = A.q_tile & 0xFFFF;
  constexpr bool fix = FIX;  // fix-up launch of a guarded plan (pcg_abi.hip): only the envs the first launch marked
  const bool nosort = (A.q_tile & 0x10000) != 0;  // measurement switch (PCG_Q_NOSORT)
  const int refill_hi = (A.q_tile >> 20) & 0x7F;  // measurement switch (PCG_Q_REFILL); 0 = the default
  // (the default depends on the tile: see where it is used)
  static_assert(QSORT == (1 << QSLOT_BITS), "slot index field");
  const bool lean = (A.q_tile & 0x40000) != 0;  // the lean LDS layout (QTile; host: it is what lets the state fit)
  const int nus = lean ? NA + c.nd : NU;
  double* us = lds;
  double* hs = us + (size_t)nus * T;           // (not in the lean layout)
  uint32_t* sortbuf = reinterpret_cast<uint32_t*>(hs + (lean ? 0 : T));
  const int Te = (T + 1) & ~1;  // (8-byte alignment of what follows)
  int32_t* accs = reinterpret_cast<int32_t*>(sortbuf + QLayout<M>::sort_words(T));  // (lean layout: both counts packed)
  int32_t* rejs = accs + Te;                                                        // (not in the lean layout)
  uint8_t* flag = reinterpret_cast<uint8_t*>(lean ? rejs : rejs + Te);   // bits 0-1: PCG_ST_* of the integration, bit 2: pre-step "done"
  int32_t* next = reinterpret_cast<int32_t*>(flag + ((T + 7) & ~7));   // queue head (the first min(n, 256) sorted slots are handed out directly)
  int32_t* nq = next + 1;     // fix-up launch: number of marked envs parked in this round
  int32_t* nheavy = next + 2; // cooperative rule (Rodas4): heavy envs of the tile = its first sorted slots ...
  int32_t* cnext = next + 3;  // ... and the head of their queue
  constexpr bool COOP = is_ros_pair(INTEG) && has_coop<M>::value && !FIX;
  const bool coop = COOP && c.coop_thr > 0.0;
  const bool xlds = (A.q_tile & 0x20000) != 0;  // the tile's state lives in LDS (host: it fits)
  double* xs = reinterpret_cast<double*>(next + 4);  // [NX][T] when xlds (8-byte aligned: every array before it is)
  int32_t* eidx = reinterpret_cast<int32_t*>(xs);  // fix-up launch: env (relative to the workgroup's range) of a compact slot
  double* sched_l = xs + (xlds ? (size_t)NX * T : fix ? (size_t)(T + 1) / 2 : 0);  // per-env-t schedule tables behind the tile
  if (PER_ENV_T) stage_schedules(A, c, sched_l);
  typename M::CKP& kp = *(typename M::CKP*)c.kp;
  const int64_t B = A.B;
  const int tid = threadIdx.x, lane = tid & 63;
  // this workgroup's contiguous range of envs, walked in sub-tiles of (almost) equal size <= T
  const int64_t per = (B + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(B, lo + per);
  if (lo >= hi) return;
  if (fix) {  // nothing marked in this workgroup's range (the calm closed loop: every launch): leave after one round of loads
    int found = 0;
    for (int64_t i = lo + tid; i < hi; i += QB) found |= A.done[i] == PCG_DONE_PENDING ? 1 : 0;
    if (!__syncthreads_or(found)) return;
  }
  const int nsub = (int)((hi - lo + T - 1) / T);
  const int64_t sub = (hi - lo + nsub - 1) / nsub;
  const double dt = c.dt, rtol = c.rtol, atol = c.atol;
#ifdef PCG_QSTATS  // per-wave record of 16 words in A.g: 0-5 wall-clock stamps (100 MHz), 6-11 the loop's counts (last sub-tile)
  unsigned long long* qst = A.g ? reinterpret_cast<unsigned long long*>(A.g) + ((size_t)blockIdx.x * (QB / 64) + (tid >> 6)) * 16 : nullptr;
#define PCG_QS(k) if (lane == 0 && qst) qst[k] = wall_clock64()
#else
  unsigned long long* qst = nullptr;
#define PCG_QS(k)
#endif
  int64_t scan = lo;  // fix-up launch: the next env of the range to look at
  for (int isub = 0; fix ? scan < hi : isub < nsub; ++isub) {
    const int64_t base = fix ? lo : lo + (int64_t)isub * sub;
    PCG_QS(0);
    if constexpr (COOP) {
      if (tid == 0) *nheavy = 0, *cnext = 0;
      __syncthreads();
    }
    int my_heavy = 0;
    // ---------------- phase 1: load, pre-integration half, park in LDS ----------------
    // one slot: the env's pre-integration half, its held input, initial step size and flags parked at slot s; returns the
    // slot's sort word
    auto park = [&](int s, int64_t e) -> uint32_t {
      const int t = PER_ENV_T ? A.t[e] : A.t_scalar;
      double x[NX], a[NA];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = A.x[(size_t)i * B + e];
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = A.a[(size_t)i * B + e];
      EnvPre<M> pre;
      env_pre<M, PER_ENV_T, EXTRAS>(A, c, sched_l, e, t, a, x, pre);
      const typename M::Hold hold = M::hold(kp, pre.u);
      const RhsFn<M> f{kp, hold};
      double k1[NX];
      double d1, h;
      bool heavy = false;
      if constexpr (is_ros_pair(INTEG)) {
        f(x, k1);
        double d0;
        h = rodas4_h_init<NX, INTEG>(x, k1, NX, dt, rtol, atol, d1, &d0);
        if constexpr (COOP) {  // the cooperative rule (the statements of seulex8_if_heavy): SEULEX-8's first big step instead
          if (coop && M::coop_key(kp, pre.u, d1) >= c.coop_thr) {
            heavy = true;
            h = seulex_h_init(d0, d1, dt);
            ++my_heavy;
          }
        }
      } else {
        h = dopri5_h_init<NX>(f, x, k1, NX, dt, rtol, atol, d1);
      }
      if (lean) {  // the action and the configured disturbance values (pre.dv); the first step is recomputed at pick-up
#pragma unroll
        for (int i = 0; i < NA; ++i) us[(size_t)i * T + s] = pre.u[i];
#pragma unroll
        for (int k = 0; k < NDM; ++k)
          if (k < c.nd) us[(size_t)(NA + k) * T + s] = pre.dv[k];
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) us[(size_t)i * T + s] = pre.u[i];
        hs[s] = h;
      }
      if (xlds) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[(size_t)i * T + s] = x[i];
      }
      flag[s] = pre.done_pre ? 4 : 0;
      float key;
      // stability-limited part (the model's rate x dt) + the initial transient's share (ln of the scaled |f(x0)|,
      // already computed for the initial step size).  A least-squares fit on BASELINE configs[2] puts the weight at
      // 33 (correlation with the measured step counts 0.84 -> 0.96, list-scheduling efficiency of independent lanes
      // 0.836 -> 0.862); lock-stepped waves prefer less: measured optimum ~20 (pcg_abi.hip, profiles/r2/queue_w_sweep.txt)
      if constexpr (is_ros_pair(INTEG))  // fitted attempts per env step of the (fourth-order) Rosenbrock pair (pcg_models.hpp)
        key = M::cost_key_ros(kp, pre.u) + A.q_w * __builtin_logf(__builtin_fmaxf((float)d1, 1.0f));
      else if constexpr (has_cost_key<M>::value)
        key = (float)(M::cost_key(kp, pre.u) * dt) + A.q_w * __builtin_logf(__builtin_fmaxf((float)d1, 1.0f));
      else key = (float)(dt / h);  // generic proxy: steps at the initial step size
      key = nosort ? 1.0f : __builtin_fmaxf(key, 1e-30f);
      // positive floats order like their bit patterns; the sign bit is free: heavy envs sort in front of all others
      return ((__float_as_uint(key) >> QSLOT_BITS) << QSLOT_BITS) | (uint32_t)s | (heavy ? 0x80000000u : 0u);
    };
    int n, S;
    if (fix) {
      // The marked envs of the range go into a COMPACT list of slots (eidx[slot] = the env), QB envs looked at per round
      // of the scan, until the range ends or the next round might not fit: a tile holds T MARKED envs, so a workgroup whose
      // share of the batch is several tiles long still starts its heaviest env first -- with sub-tiles of the range the
      // second one's ignition front (~130 attempts of a lone lane) only started when the first one's had finished.
      if (tid == 0) *nq = 0;
      __syncthreads();
      int parked = 0;
      while (scan < hi && parked + QB <= T) {
        const int64_t e = scan + tid;
        if (e < hi && A.done[e] == PCG_DONE_PENDING) {
          const int s = atomicAdd(nq, 1);
          eidx[s] = (int32_t)(e - lo);
          sortbuf[s] = park(s, e);
        }
        __syncthreads();
        parked = *nq;
        __syncthreads();  // (every thread has read the count before the next round adds to it)
        scan += QB;
      }
      n = parked;
      if (n == 0) continue;  // (uniform)
      S = n <= QSORT / 4 ? QSORT / 4 : (n <= QSORT / 2 ? QSORT / 2 : QSORT);
      for (int s = n + tid; s < S; s += QB) sortbuf[s] = (uint32_t)s;  // padding: sorts behind every real slot
    } else {
      n = (int)(min(hi, base + sub) - base);  // envs in this sub-tile
      S = n <= QSORT / 4 ? QSORT / 4 : (n <= QSORT / 2 ? QSORT / 2 : QSORT);  // sort width
      for (int s = tid; s < S; s += QB) sortbuf[s] = s < n ? park(s, base + s) : (uint32_t)s;
    }
    PCG_QS(1);
    if constexpr (COOP) {
      if (my_heavy > 0) atomicAdd(nheavy, my_heavy);
    }
    __syncthreads();
    // (ordered before phase 2 by the sort's barriers).  With heavy envs in the tile the pair's queue starts behind them and
    // hands nothing out directly (queue_integrate)
    if (tid == 0) *next = (COOP && *nheavy > 0) ? *nheavy : (QB < n ? QB : n);
    // ---------------- sort the slots by decreasing cost key ----------------
    if (S == QSORT / 4) sort_tile<QSORT / 4 / QB, QB>(sortbuf);
    else if (S == QSORT / 2) sort_tile<QSORT / 2 / QB, QB>(sortbuf);
    else sort_tile<QSORT / QB, QB>(sortbuf);
#ifdef PCG_QSTATS  // the results do not depend on the order: only a probe can tell whether the sort sorts
    {
      int bad = 0;
      for (int i = tid; i + 1 < S; i += QB) bad += sortbuf[i] < sortbuf[i + 1] ? 1 : 0;
      const unsigned long long bb = __ballot(bad > 0);
      if (lane == 0 && qst) qst[12] = __popcll(bb);
    }
#endif
    // ---------------- phase 2: the work queue ----------------
    // idle lanes that trigger a refill: with at most two envs per lane every lane refills once and waiting for company
    // only idles it (me10: 0.678 ms at 8, 0.656 at 2); with more envs per lane the refill code -- executed by the whole
    // wave -- is worth batching (configs[4] shard: 0.916 ms at 8, 0.929 at 2; profiles/r2/queue_refill_sweep.txt)
    const int refill = refill_hi ? refill_hi : (n <= 2 * QB ? 2 : QREFILL);
    PCG_QS(2);
    const QTile<M> Q{us, lean ? nullptr : hs, accs, lean ? nullptr : rejs, flag, T, lean, c.nd, c.d_slot, c.d_default};
    int nh = 0;
    if constexpr (COOP) {
      nh = *nheavy;  // (uniform; written before the sort's barriers)
      // how many of the workgroup's waves take part (0 = all): with few heavy envs per tile every wave would carry a mostly
      // empty set of groups through the phase -- one wave with all eight groups busy costs the tile less
      const int coop_waves = (A.q_tile >> 27) & 0xF;
      if (nh > 0 && (coop_waves == 0 || (tid >> 6) < coop_waves)) {
        if (A.q_prio > 0) __builtin_amdgcn_s_setprio(3);  // the heavy envs are the launch's critical path
        coop_integrate<M, QB>(&kp, xlds ? xs : A.x + base, xlds ? (int64_t)T : B, Q, sortbuf, cnext, nh, dt,
                              c.dt_edge, c.h_floor, rtol, atol, c.max_steps, c.ep_c, c.ep_kmax, qst);
        if (A.q_prio > 0) __builtin_amdgcn_s_setprio(0);
      }
    }
    PCG_QS(13);
    queue_integrate<M, INTEG, QB, FIX>(&kp, xlds ? xs : A.x + base, xlds ? (int64_t)T : B, Q, sortbuf, next, n, refill, dt, c.dt_edge, c.h_floor, rtol, atol, c.max_steps, c.ep_c, c.ep_kmax, A.q_prio,
                              (A.q_prio > 0 && blockIdx.x >= gridDim.x / 2) ? 2 * 64 : 0, qst, fix ? eidx : nullptr, nh);
    if (A.q_prio > 0) __builtin_amdgcn_s_setprio(0);
    PCG_QS(3);
    __syncthreads();
    PCG_QS(4);
    // ---------------- phase 3: post-integration half, coalesced stores ----------------
    for (int s = tid; s < n; s += QB) {
      const int64_t e = fix ? lo + eidx[s] : base + s;
      const int t = PER_ENV_T ? A.t[e] : A.t_scalar;
      double x[NX];
      EnvPre<M> pre;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = xlds ? xs[(size_t)i * T + s] : A.x[(size_t)i * B + e];  // written by phase 2 (same workgroup, behind a barrier)
      Q.load_u(s, pre.u);
#pragma unroll
      for (int k = 0; k < PCG_MAX_NDM; ++k) pre.dv[k] = 0.0;
#pragma unroll
      for (int k = 0; k < NDM; ++k)
        if (k < c.nd) {  // the configured disturbance values are the matching entries of the held input
          const int dslot = c.d_slot[k];
          double v = 0.0;
#pragma unroll
          for (int j = 0; j < NDM; ++j) v = (j == dslot) ? pre.u[NA + j] : v;
          pre.dv[k] = v;
        }
      const int fl = flag[s];
      pre.done_pre = (fl & 4) != 0;
      if (A.nsteps) {
        int na_, nr_;
        Q.counts(s, na_, nr_);
        A.nsteps[e] = na_;
        A.nsteps[B + e] = nr_;
      }
      EnvOut<M> out;
      env_post<M, PER_ENV_T, EXTRAS>(A, c, sched_l, e, t, pre, x, finite_status<NX>(fl & 3, x, NX), out);
      if (A.auto_reset && out.done) {
        __builtin_nontemporal_store(out.rew, A.rew + e);
        A.done[e] = 1;
        if (A.viol) A.viol[e] = out.viol ? 1 : 0;
        if (A.status && out.status != PCG_ST_OK) A.status[e] = out.status;
        reset_env(A, c, e, A.reset_seed);
        continue;
      }
      if (xlds) {  // the new state goes back to the batch, coalesced (otherwise phase 2 has put it in place)
#pragma unroll
        for (int i = 0; i < NX; ++i) A.x[(size_t)i * B + e] = x[i];
      }
      store_out<M>(A, c, e, out, A.obs + e);
      if (PER_ENV_T) A.t[e] = t + 1;
    }
    PCG_QS(5);
    __syncthreads();  // the tile's LDS is reused by the next sub-tile
  }
}

}  // namespace pcg
