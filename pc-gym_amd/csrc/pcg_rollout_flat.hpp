// pcg_rollout_flat.hpp -- barrier-free fused rollout of a guarded plan (PCG_INT_T5G, the cstr's default), second pass.
//
// Why.  Env i at step t + 1 needs only env i at step t -- but every pcg_step launch is a batch-wide barrier, and the
// reference's CVODES (integrator.py:163-182, called per env at pcgym.py:423-429) never pays for its batch-mates.  Under the
// guarded plan an env is either CALM (the fixed step is trusted: one cheap step) or HOT (ignited: the adaptive pair, ~4
// attempts per step, ~100 across the ignition front), and measured on the oracle over a 59-step episode of the full x0 box
// (tools/barrier_probe.py, profiles/r6/barrier_probe.txt): hot is absorbing (P(hot at t + 1 | hot at t) = 0.993, P(hot | calm)
// = 0.000 per step), the sum over steps of the batch's heaviest env is 3980 attempts where the heaviest env's own episode
// is 545 (7.3 x), and a wave that keeps 64 envs in lock step pays 1346 where its lanes need 106 on average.
//
// How.  Two launches per rollout instead of one (or T):
//   pass 1  rollout_kernel<M, PCG_INT_T5G> with StepArgs::fixup set: one env per lane, step after step while the guard
//           trusts the fixed step; the first step it does not trust ends the env's part in this pass -- its state is the
//           step's start state, flat_tstar[e] the step, and e goes onto the hand-over list flat_hot.  Coalesced, HBM-side.
//   pass 2  rollout_kernel_hot<M> (below): persistent waves, every LANE carries one handed-over env from its step t* to the
//           end of the rollout and then pulls the next one from the list (one wave-aggregated atomic per refill).  Per loop
//           iteration a lane is in one phase -- START (action, pre-step half, the guard at the start state), FIX (the rare
//           hot env whose start guard holds: the whole guarded fixed step), ADAPT (ONE attempt of the adaptive pair), POST
//           (post-step half, outputs of step s at their place in the trajectories) -- so a lane crossing an ignition front
//           holds up nobody: its neighbours finish their steps, their episodes, and pull new envs meanwhile.
// The start-state guard decides alone that a step is NOT trusted (calm and slow only ever go from true to false inside
// t5_guarded): a hot env skips the six further right-hand sides of a fixed step whose result it would throw away.
//
// Same statements per env step as stepping (env_pre, t5_guarded, dopri5_h_init + dopri5_attempt -- the resumable form of
// dopri5() that the work-queue kernels use --, poison_if_failed, finite_status, env_post): the trajectories are bitwise those
// of T pcg_step launches (tests/test_gpu_flat_rollout.py), whatever the order the lanes happen to finish in.
// Not with a_delta (env_pre accumulates into a_save: the first pass would apply a handed-over step's action twice).
#pragma once

namespace pcg {

constexpr int FLAT_BLOCK = 256;
constexpr int FLAT_REFILL = 8;  // idle lanes that trigger a pull from the hand-over list (or: no busy lane left)

#ifndef PCG_FLAT_WPE
#define PCG_FLAT_WPE 1  // min waves per SIMD asked of the register allocator (A/B builds)
#endif
#ifndef PCG_FLAT_ATT
#define PCG_FLAT_ATT 2  // attempts of the adaptive pair per loop iteration (A/B builds)
#endif
template <class M>
__global__ __launch_bounds__(FLAT_BLOCK, PCG_FLAT_WPE) void rollout_kernel_hot(const StepArgs A) {
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  const int64_t B = A.B;
  const int T = A.T;
  typename M::CKP& kp = model_kp<M>(c);
  const int n_items = A.flat_q[1];  // written by the first pass (an earlier launch on the same stream)
  int32_t* head = A.flat_q + 2;
  enum { IDLE = 0, START = 1, FIX = 2, ADAPT = 3, POST = 4 };
  int64_t e = -1;
  int s = 0, phase = IDLE, nacc = 0, nrej = 0, status = PCG_ST_OK;
  double x[NX], a_nxt[NA];  // a_nxt: the action of step s, in flight since the previous step's START
  EnvPre<M> pre;
  DpLane<NX> L;
  bool drained = n_items <= 0;  // wave-uniform
  const int every = A.q_tile > 0 ? A.q_tile : 1;  // the step boundaries run every that many iterations (host: FLAT_EVERY)
  // every spin is bounded (max_steps bounds each env step): a logic error becomes a flagged result, not a hung GPU
  const long long cap64 = ((long long)c.max_steps + 8) * (long long)T * 4 + 64;
  for (long long iter = 0; iter < cap64 * 64; ++iter) {
    const bool busy = e >= 0;
    const unsigned long long bm = __ballot(busy);
    const int n_idle = __popcll(__ballot(true)) - __popcll(bm);
    if (!drained && (n_idle >= FLAT_REFILL || bm == 0ull)) {
      const unsigned long long im = __ballot(!busy);
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
      int got = 0;
      if (!busy && rank == 0) got = atomicAdd(head, n_idle);
      got = __shfl(got, __ffsll((long long)im) - 1);
      drained = got + n_idle >= n_items;
      if (!busy) {
        const int j = got + rank;
        if (j < n_items) {
          e = A.flat_hot[j];
          s = A.flat_tstar[e];
#pragma unroll
          for (int i = 0; i < NX; ++i) x[i] = A.x[(size_t)i * B + e];
#pragma unroll
          for (int i = 0; i < NA; ++i) a_nxt[i] = A.a_seq[(size_t)s * A.a_ss + (size_t)i * A.a_cs + e];
          phase = START;
        }
      }
    }
    if (__ballot(e >= 0) == 0ull) break;
    if (phase == ADAPT) {  // one attempted step of the adaptive pair
      const typename M::Hold hold = M::hold(kp, pre.u);
      const RhsFn<M> f{kp, hold};
      int st = dopri5_attempt<NX>(f, L, NX, c.dt, c.dt_edge, c.h_floor, c.rtol, c.atol, c.max_steps);
#pragma unroll
      for (int r = 1; r < PCG_FLAT_ATT; ++r)
        if (st < 0) st = dopri5_attempt<NX>(f, L, NX, c.dt, c.dt_edge, c.h_floor, c.rtol, c.atol, c.max_steps);
      if (st >= 0) {
        poison_if_failed<NX>(st, L.x);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = L.x[i];
        nacc = L.acc, nrej = L.rej, status = st;
        phase = POST;
      }
    }
    // The step boundaries (POST of the finished step, START of the next one) cost the wave more instructions than an attempt
    // of this two-state model (~650 against ~550), whichever lanes need them: they run every `every`-th iteration -- a lane that
    // finishes in between waits -- or at once when no lane of the wave is mid-step (measured: profiles/r6/flat_rollout.txt).
    const bool boundary = (iter % every) == 0 || __ballot(phase == ADAPT) == 0ull;
    if (!boundary) continue;
    if (phase == POST) {  // the post-step half and the outputs of step s, each at its own place in the trajectories
      if (A.nsteps) {
        A.nsteps[e] = nacc;
        A.nsteps[B + e] = nrej;
      }
      EnvOut<M> out;
      env_post<M, false, true, false>(A, c, nullptr, e, A.t_scalar + s, pre, x, finite_status<NX>(status, x, NX), out);
      if (A.rew_seq) A.rew_seq[(size_t)s * A.r_ss + e] = out.rew;
      if (A.obs_seq) store_obs<M, false>(A, c, out, A.obs_seq + (size_t)s * A.o_ss + e, A.o_cs);
      if (s == T - 1 || !A.obs_seq) store_out<M, false>(A, c, e, out, A.obs + e);
      ++s;
      if (s == T) {
#pragma unroll
        for (int i = 0; i < NX; ++i) A.x[(size_t)i * B + e] = x[i];
        e = -1;
        phase = IDLE;
      } else {
        phase = START;
      }
    }
    if (phase == START) {  // the step's action, the pre-step half, and what the guard says about the start state
      double a[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = a_nxt[i];
      if (s + 1 < T) {  // the next step's action travels while this step is integrated (a scattered 8-byte load per lane)
        const double* as = A.a_seq + (size_t)(s + 1) * A.a_ss;
#pragma unroll
        for (int i = 0; i < NA; ++i) a_nxt[i] = as[(size_t)i * A.a_cs + e];
      }
      env_pre<M, false, true>(A, c, nullptr, e, A.t_scalar + s, a, x, pre);
      const typename M::Hold hold = M::hold(kp, pre.u);
      double g, rho;
#pragma unroll
      for (int i = 0; i < NX; ++i) L.x[i] = x[i];
      M::rhs_guard(kp, hold, L.x, L.k1, g, rho);  // k1 = f(x): the same bits as rhs() (pcg_models.hpp)
      bool calm[1] = {true}, slow[1] = {true};
      guard_acc<double, 1>(g, rho, c.h, T5G_SLOW_LIMIT, calm, slow);
      if (calm[0] && slow[0]) {
        phase = FIX;
      } else {  // not trusted whatever the fixed step would give: the adaptive pair from the start state
        const RhsFn<M> f{kp, hold};
        L.t = 0.0;
        L.acc = L.rej = 0;
        L.rejected_last = false;
        L.h = dopri5_h_init_k1<NX>(f, L.x, L.k1, NX, c.dt, c.rtol, c.atol);
        phase = ADAPT;
      }
    }
    if (phase == FIX) {  // (rare in this pass: an env that was handed over and whose start guard holds again)
      const typename M::Hold hold = M::hold(kp, pre.u);
      double x0[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) x0[i] = x[i];
      int g1[1];
      t5_guarded<M, typename M::CKP, double>(kp, hold, x, c.h, c.substeps, g1);
      if (g1[0] == 0) {
        nacc = nrej = 0, status = PCG_ST_OK;
        phase = POST;
      } else {
        const RhsFn<M> f{kp, hold};
#pragma unroll
        for (int i = 0; i < NX; ++i) L.x[i] = x[i] = x0[i];
        L.t = 0.0;
        L.acc = L.rej = 0;
        L.rejected_last = false;
        double d1;
        L.h = dopri5_h_init<NX>(f, L.x, L.k1, NX, c.dt, c.rtol, c.atol, d1);
        phase = ADAPT;
      }
    }
  }
  if (e >= 0) {  // (not reached by a correct run: the iteration bound ended the loop with this env in flight)
#pragma unroll
    for (int i = 0; i < NX; ++i) A.x[(size_t)i * B + e] = __builtin_nan("");
    if (A.status) A.status[e] = PCG_ST_MAX_STEPS;
  }
}

}  // namespace pcg
