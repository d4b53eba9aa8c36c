// pcg_step_feat.hpp -- the general env step on the lean kernel's execution shape.
//
// Round 1 had two step paths: the lean kernel (two envs per lane as one Pack<2> value, persistent grid, next tile's
// inputs in flight while the current one is integrated: 0.79 of the HBM peak) and a general kernel for everything a
// configuration can switch on (noise, Gaussian / per-env disturbances, constraint rows, a_delta, tracking / terminal
// rewards, per-env step counters, same-launch auto-reset) -- one env per lane, every feature a RUN-TIME branch on
// cfg.flags: 0.43-0.51 of the peak, 340-760 VALU + 200-370 SALU wave-instructions per env against 207 + 55
// (profiles/r1/exploration.md).  This header is that general step re-written on the lean shape:
//
//   * env_step_feat<M, W, FT>: W envs per lane through one instruction stream (independent dependency chains
//     interleave), every optional feature behind a COMPILE-TIME bit of the mask FT -- a feature that is not in the
//     mask costs neither scalar branches nor registers; a feature in the mask is still gated by cfg.flags, so an
//     instantiation serves every configuration whose needs are a SUBSET of its mask;
//   * step_kernel_feat<M, W, FT>: the software-pipelined persistent kernel of step_kernel_pipe with the feature's
//     extra per-env rows (a_delta accumulator, previous action) riding in the same prefetch, 16 bytes per lane and row;
//   * feat_table<M>(): the curated list of masks built for a model (the single features, the pairs the paper
//     configurations use, "everything"); the host picks the smallest superset of what a launch needs.
//
// Measured (profiles/r2/extras_probe*.txt, cstr B = 2^20, two boxes): this shape wins for constraint rows (15.7 vs 18.3
// and 17.5 vs 18.6 us), ties or wins for the tracking reward (16.0 vs 17.4, 16.9 vs 16.9), and does NOT beat the classic
// one-env-per-lane kernel for observation noise (16.1 vs 16.5 on one box, 17.4 vs 16.8 on the other; with tracking
// 19.3-20.3 vs 18.9), per-env step counters (16.0 vs 14.9) and Gaussian / per-env disturbances (18.0 vs 17.2) -- so
// those stay on the classic kernel and are not features here (the first cuts had them: removed with the data).
//
// Statement order follows make_env.step (reference src/pcgym/pcgym.py:350-500) exactly as env_step does; both are
// checked against the same oracle recordings.  Only RK4 plans of the small HBM-bound models (NX <= 4) come here:
// the adaptive and the wide models are bound by fp64 issue, not by how the epilogue is compiled.
#pragma once

namespace pcg {

enum : unsigned {
  FT_CONS = 4u,     // constraint rows, penalty, done-on-violation   pcgym.py:414-420, 443-446, 560-615
  FT_ADELTA = 8u,   // PCG_F_A_DELTA                    pcgym.py:376-383
  FT_TRACK = 16u,   // PCG_F_REWARD_TRACK               pc-gym_paper/.../custom_reward.py
  FT_BATCH = 32u,   // PCG_F_REWARD_BATCH               pcgym.py:502-532
  FT_AR = 256u,     // same-launch auto-reset of a lock-stepped batch (pcg_step_autoreset)
  FT_ALL = FT_CONS | FT_ADELTA | FT_TRACK | FT_BATCH | FT_AR
};

template <int W>
PCG_DEV Pack<W> pk_clamp(const Pack<W>& a, double lo, double hi) {
  Pack<W> r;
#pragma unroll
  for (int j = 0; j < W; ++j) r.v[j] = fmin(fmax(a.v[j], lo), hi);
  return r;
}
// what one tile's lane holds on arrival: the W envs' state and action, plus the rows the features need
template <class M, int W, unsigned FT>
struct FeatIn {
  using V = typename Vec<W>::T;
  V x[M::NX], a[M::NA], asave[M::NA], uprev[M::NA];
};

template <class M, int W>
struct FeatOut {
  static constexpr int ND = M::NDM > 0 ? M::NDM : 1;
  Pack<W> ox[M::NX], rew, asave[M::NA];
  double osp[PCG_MAX_NSP], od[ND];  // wave-uniform (lock-stepped batch, shared schedules)
  bool done[W], viol[W];
  uint8_t status[W];
  bool reset;  // this lane's envs were reset in the launch (FT_AR)
};

// constraint rows g = A.[x|sp|d|u] - b for the W envs of a lane (affine form of the reference's callable,
// pcgym.py:560-577); rows go to gout (16-byte store per row) when it is non-null.
template <class M, int W>
PCG_DEV void constraint_rows_w(CDevConst& c, const Pack<W> (&x)[M::NX], const double (&spv)[PCG_MAX_NSP],
                               const double (&dv)[FeatOut<M, W>::ND], const Pack<W> (&u)[M::NA + M::NDM], double* gout,
                               int64_t B, int64_t e0, bool (&violated)[W]) {
#pragma unroll
  for (int j = 0; j < W; ++j) violated[j] = false;
  for (int r = 0; r < c.ncon; ++r) {
    const PCG_CONSTANT double* row = c.con_A[r];
    double gu = -c.con_b[r];  // the wave-uniform part of the row first (scalar unit)
#pragma unroll
    for (int k = 0; k < PCG_MAX_NSP; ++k) gu += row[PCG_MAX_NX + k] * spv[k];
#pragma unroll
    for (int k = 0; k < M::NDM; ++k) gu += row[PCG_MAX_NX + PCG_MAX_NSP + k] * dv[k];
    Pack<W> g(gu);
#pragma unroll
    for (int i = 0; i < M::NX; ++i) g = g + row[i] * x[i];
#pragma unroll
    for (int j = 0; j < M::NA; ++j) g = g + row[PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + j] * u[j];
#pragma unroll
    for (int j = 0; j < M::NDM; ++j)
      g = g + row[PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + PCG_MAX_NA + j] * u[M::NA + j];
    if (gout) Vec<W>::store_nt(gout + (size_t)r * B + e0, g.v);
#pragma unroll
    for (int j = 0; j < W; ++j) violated[j] |= (g.v[j] > 0.0);
  }
}

// initial state of one env (pcgym.py:284-288, apply_uncertainties :255-261) and its observation rows: the draws of
// reset_env / reset_lean, kept in registers so that the caller can merge them into a vector store
template <class M>
PCG_DEV void reset_vals(const StepArgs& A, CDevConst& c, uint64_t env_id, uint64_t seed, double (&xv)[M::NX],
                        double (&ov)[M::NX]) {
#pragma unroll
  for (int i = 0; i < M::NX; ++i) {
    double v = c.x0[i];
    if (c.has_x0_unc && c.x0_unc[i] != 0.0) {
      const double pct = c.x0_unc[i];
      if (c.flags & PCG_F_X0_NORMAL) {
        double z0, z1;
        rng_normal2(seed, env_id, 0u, RNG_RESET + (uint32_t)(i >> 1), z0, z1);
        v = c.x0[i] + pct * c.x0[i] * ((i & 1) ? z1 : z0);
      } else {
        double u0, u1;
        rng_uniform2(seed, env_id, 0u, RNG_RESET + (uint32_t)(i >> 1), u0, u1);
        v = c.x0[i] * (1 + pct * (2.0 * ((i & 1) ? u1 : u0) - 1.0));
      }
    }
    xv[i] = v;
    ov[i] = (v - c.omap[i].lo) * c.omap[i].sc + c.omap[i].off;
  }
}

// ---------------------------------------------------------------------------
// One env step for the W envs of a lane, features by compile-time mask.  The batch is lock-stepped: the step
// counter, the schedule values, the SP / disturbance slots are wave-uniform scalars.
// ---------------------------------------------------------------------------
template <class M, int W, unsigned FT>
PCG_DEV void env_step_feat(const StepArgs& A, CDevConst& c, int64_t e0, int t, const Pack<W> (&a_in)[M::NA],
                           const Pack<W> (&asave_in)[M::NA], const Pack<W> (&uprev_in)[M::NA], Pack<W> (&x)[M::NX],
                           FeatOut<M, W>& out) {
  static_assert(!M::DYNAMIC, "the feature kernels are built for the fixed-size models");
  constexpr int NX = M::NX, NA = M::NA, NDM = M::NDM, ND = FeatOut<M, W>::ND;
  constexpr bool CONS = FT & FT_CONS, ADELTA = FT & FT_ADELTA, TRACK = FT & FT_TRACK,
                 BATCH = FT & FT_BATCH, AR = FT & FT_AR;
  using R = Pack<W>;
  const int64_t B = A.B;
  const uint32_t flags = c.flags;
  const int N = c.N, nsp = c.nsp, nso = c.nsp_obs, nd = c.nd;
  typename M::CKP& kp = *(typename M::CKP*)c.kp;
  const int tn = min(t + 1, N - 1);  // schedule index clamp (the reference would IndexError)
  const int tc = min(t, N - 1);
  // ---- action map (pcgym.py:371-383) ----
  R u[NA + NDM];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    R av = action_map(a_in[i], c.a_lo[i], c.a_hi[i], (flags & PCG_F_NORMALISE_A) != 0,
                      (flags & PCG_F_A_DELTA) && (flags & PCG_F_REF_COMPAT));
    if constexpr (ADELTA) {
      if (flags & PCG_F_A_DELTA) {
        av = asave_in[i] + av;  // Q2: the unclipped sum drives the plant
        out.asave[i] = pk_clamp(av, c.a_act_lo[i], c.a_act_hi[i]);
      }
    }
    u[i] = av;
  }
  // ---- disturbance injection from the shared schedule (pcgym.py:386-412) ----
  double dv[ND], ud[ND];
#pragma unroll
  for (int k = 0; k < ND; ++k) dv[k] = 0.0;
#pragma unroll
  for (int j = 0; j < NDM; ++j) ud[j] = c.d_default[j];
#pragma unroll
  for (int k = 0; k < NDM; ++k)
    if (k < nd) {
      const double v = A.sched[(size_t)(nsp + k) * N + tn];  // Q6: index t+1
      dv[k] = v;
      const int slot = c.d_slot[k];
#pragma unroll
      for (int j = 0; j < NDM; ++j) ud[j] = (j == slot) ? v : ud[j];
    }
#pragma unroll
  for (int j = 0; j < NDM; ++j) u[NA + j] = R(ud[j]);
  // ---- pre-step constraint check at t == 0 (pcgym.py:414-420) ----
  bool done[W], violated[W];
#pragma unroll
  for (int j = 0; j < W; ++j) done[j] = violated[j] = false;
  if constexpr (CONS) {
    if (c.ncon > 0 && t == 0) {
      double sp0[PCG_MAX_NSP];
#pragma unroll
      for (int k = 0; k < PCG_MAX_NSP; ++k) sp0[k] = (k < nso) ? c.x0[NX + k] : 0.0;
      bool v0[W];
      constraint_rows_w<M, W>(c, x, sp0, dv, u, A.g_pre, B, e0, v0);
#pragma unroll
      for (int j = 0; j < W; ++j) done[j] = v0[j] && (flags & PCG_F_DONE_ON_CONS);
    }
  }
  // ---- integrate over [0, dt], u held (pcgym.py:423-429, integrator.py:90-107,163-182) ----
  {
    const typename M::template HoldT<R> hold = M::template hold<R>(kp, u);
    const RhsFn<M, R> f{kp, hold};
    rk4<NX>(f, x, c.h, c.h2, c.h6, c.substeps);
  }
  // ---- SP slot uses SP[t_old] (pcgym.py:432-438, quirk Q5); t += 1 ----
  double spv[PCG_MAX_NSP], spn[PCG_MAX_NSP];
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k) {
    spv[k] = (k < nsp) ? A.sched[(size_t)k * N + tc] : 0.0;
    spn[k] = (k < nsp) ? A.sched[(size_t)k * N + tn] : 0.0;
  }
  const int t_new = t + 1;
  // ---- post-step constraints (pcgym.py:443-446) ----
  if constexpr (CONS) {
    if (c.ncon > 0) {
      constraint_rows_w<M, W>(c, x, spv, dv, u, A.g, B, e0, violated);
#pragma unroll
      for (int j = 0; j < W; ++j) done[j] |= violated[j] && (flags & PCG_F_DONE_ON_CONS);
    }
  }
#pragma unroll
  for (int j = 0; j < W; ++j) {
    done[j] |= (t_new == N - 1);  // pcgym.py:448-449
    out.done[j] = done[j];
    out.viol[j] = violated[j];
  }
  // ---- reward on the noise-free state (pcgym.py:470-482) ----
  R r(0.0);
  bool batch = false;
  if constexpr (BATCH) batch = (flags & PCG_F_REWARD_BATCH) != 0;
  if (batch) {  // pcgym.py:502-532
    if (t_new == N - 1) {
      for (int k = 0; k < c.nrew; ++k) {
        const R v = pick<NX, W>(x, c.rew_index[k]) * c.r_scale[k];
        r = (flags & PCG_F_MAXIMISE) ? r + v : r - v;
      }
      if constexpr (CONS) {
        if (flags & PCG_F_R_PENALTY) {
#pragma unroll
          for (int j = 0; j < W; ++j) r.v[j] -= violated[j] ? 1000.0 : 0.0;
        }
      }
    }
  } else {  // pcgym.py:535-558
#pragma unroll
    for (int k = 0; k < PCG_MAX_NSP; ++k)
      if (k < nsp) {
        const R dd = pick<NX, W>(x, c.sp_index[k]) - spn[k];
        r = r + (-(dd * dd)) * c.r_scale[k];
        if constexpr (CONS) {
          if (flags & PCG_F_R_PENALTY) {
#pragma unroll
            for (int j = 0; j < W; ++j) r.v[j] -= violated[j] ? 1000.0 : 0.0;  // Q4: once per SP key
          }
        }
      }
  }
  // ---- observation (no noise on this shape: measured slower than the classic kernel): normalise (pcgym.py:483-489),
  //      mask (:495-498) ----
  R on[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) on[i] = x[i];
#pragma unroll
  for (int i = 0; i < NX; ++i) out.ox[i] = (on[i] - c.omap[i].lo) * c.omap[i].sc + c.omap[i].off;
  if constexpr (TRACK) {
    if (flags & PCG_F_REWARD_TRACK) {
      // declarative form of the paper's custom_reward family (custom_reward.py:3-39; constraint_showcase/
      // custom_reward.py:6-69) on the (noisy) physical observation, as in env_step
      R cost(0.0);
#pragma unroll
      for (int k = 0; k < PCG_MAX_NSP; ++k)
        if (k < nsp) {
          const R xn = (pick<NX, W>(on, c.sp_index[k]) - c.trk_lo[k]) * c.trk_inv[k];
          const double sn = (spn[k] - c.trk_lo[k]) * c.trk_inv[k];
          cost = cost + ((xn - sn) * (xn - sn)) * c.r_scale[k];
        }
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        R up;
#pragma unroll
        for (int j = 0; j < W; ++j)  // NaN: no previous action yet (hasattr branch, custom_reward.py:7-8)
          up.v[j] = (uprev_in[q].v[j] == uprev_in[q].v[j]) ? uprev_in[q].v[j] : u[q].v[j];
        const R un = (u[q] - c.act_lo[q]) * c.act_inv[q];
        const R upn = (up - c.act_lo[q]) * c.act_inv[q];
        cost = cost + c.R_du * ((un - upn) * (un - upn)) + c.R_u * (un * un);
        *reinterpret_cast<typename Vec<W>::T*>(A.u_prev + (size_t)q * B + e0) = Vec<W>::make(u[q].v);
      }
      if constexpr (CONS) {
        for (int q = 0; q < c.nbox; ++q) {
          const R xn = (pick<NX, W>(on, c.box_index[q]) - c.box_lo[q]) * c.box_inv[q];
#pragma unroll
          for (int j = 0; j < W; ++j)
            if (violated[j]) {
              if (xn.v[j] > c.box_hin[q]) cost.v[j] += (xn.v[j] - c.box_hin[q]) * (xn.v[j] - c.box_hin[q]);
              else if (xn.v[j] < c.box_lon[q]) cost.v[j] += (c.box_lon[q] - xn.v[j]) * (c.box_lon[q] - xn.v[j]);
            }
        }
      }
      r = -cost;
    }
  }
  out.rew = r;
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nso) out.osp[k] = (spv[k] - c.omap[NX + k].lo) * c.omap[NX + k].sc + c.omap[NX + k].off;
#pragma unroll
  for (int k = 0; k < NDM; ++k)
    if (k < nd) out.od[k] = (dv[k] - c.omap[NX + nso + k].lo) * c.omap[NX + nso + k].sc + c.omap[NX + nso + k].off;
  // ---- per-env health (PCG_ST_*): fixed-step RK4 cannot fail in the integrator, only leave a non-finite state ----
#pragma unroll
  for (int j = 0; j < W; ++j) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NX; ++i) ok = ok && (__builtin_fabs(x[i].v[j]) < __builtin_inf());
    out.status[j] = ok ? PCG_ST_OK : PCG_ST_NONFINITE;
  }
  // ---- same-launch auto-reset (pcg_step_autoreset): reward / done / viol / status of the finished step stay,
  //      state, observation and a_delta accumulator become those of the new episode ----
  out.reset = false;
  if constexpr (AR) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < W; ++j) any |= done[j];
    if (A.auto_reset && any) {
      out.reset = true;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        double xv[NX], ov[NX];
        reset_vals<M>(A, c, (uint64_t)(A.env_offset + e0 + j), A.reset_seed, xv, ov);
        if (done[j]) {
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            x[i].v[j] = xv[i];
            out.ox[i].v[j] = ov[i];
          }
          if constexpr (ADELTA) {
#pragma unroll
            for (int i = 0; i < NA; ++i) out.asave[i].v[j] = c.a_0[i];
          }
        }
      }
      // the SP / disturbance slots of a reset observation (pcgym.py:291-298, quirk Q6: disturbances[k][0]); a
      // lock-stepped batch without done-on-violation ends for every env at once, so they stay wave-uniform
#pragma unroll
      for (int k = 0; k < PCG_MAX_NSP; ++k)
        if (k < nso) out.osp[k] = (c.x0[NX + k] - c.omap[NX + k].lo) * c.omap[NX + k].sc + c.omap[NX + k].off;
#pragma unroll
      for (int k = 0; k < NDM; ++k)
        if (k < nd) {
          const int q = NX + nso + k;
          out.od[k] = (A.sched[(size_t)(nsp + k) * N] - c.omap[q].lo) * c.omap[q].sc + c.omap[q].off;
        }
    }
  }
}

template <class M, int W, unsigned FT>
PCG_DEV void store_feat(const StepArgs& A, CDevConst& c, int64_t e0, const Pack<W> (&x)[M::NX],
                        const FeatOut<M, W>& out, bool nt) {
  using V = typename Vec<W>::T;
  constexpr int NX = M::NX;
  const int64_t B = A.B;
  const int nso = c.nsp_obs, nd = c.nd;
  auto put = [&](double* p, const double(&v)[W]) {
    if (nt) Vec<W>::store_nt(p, v);
    else *reinterpret_cast<V*>(p) = Vec<W>::make(v);
  };
  double tmp[W];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    *reinterpret_cast<V*>(A.x + (size_t)i * B + e0) = Vec<W>::make(x[i].v);
    put(A.obs + (size_t)i * B + e0, out.ox[i].v);
  }
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nso) {
#pragma unroll
      for (int j = 0; j < W; ++j) tmp[j] = out.osp[k];
      put(A.obs + (size_t)(NX + k) * B + e0, tmp);
    }
#pragma unroll
  for (int k = 0; k < M::NDM; ++k)
    if (k < nd) {
#pragma unroll
      for (int j = 0; j < W; ++j) tmp[j] = out.od[k];
      put(A.obs + (size_t)(NX + nso + k) * B + e0, tmp);
    }
  put(A.rew + e0, out.rew.v);
  auto store8 = [&](uint8_t* p, const uint8_t(&b)[W]) {
    if (W == 2) *reinterpret_cast<uint16_t*>(p + e0) = (uint16_t)((uint32_t)b[0] | ((uint32_t)b[W - 1] << 8));
    else p[e0] = b[0];
  };
  uint8_t bd[W], bv[W];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    bd[j] = out.done[j] ? 1 : 0;
    bv[j] = out.viol[j] ? 1 : 0;
  }
  store8(A.done, bd);
  if (A.viol) store8(A.viol, bv);
  if (A.status) {  // sticky: only failures are written
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (out.status[j] != PCG_ST_OK) A.status[e0 + j] = out.status[j];
  }
  if constexpr ((FT & FT_ADELTA) != 0) {
    if (c.flags & PCG_F_A_DELTA) {
#pragma unroll
      for (int i = 0; i < M::NA; ++i) *reinterpret_cast<V*>(A.a_save + (size_t)i * B + e0) = Vec<W>::make(out.asave[i].v);
    }
  }
}

// ---------------------------------------------------------------------------
// The persistent software-pipelined step kernel (see step_kernel_pipe for the loop structure and the reason for
// the placement of land()): each wave walks over its tiles of 256*W envs and always has the NEXT tile's inputs in
// flight while it integrates the current one.
// ---------------------------------------------------------------------------
template <class M, int W, unsigned FT>
__global__ __launch_bounds__(BLOCK, 1) void step_kernel_feat(const StepArgs A) {
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  constexpr bool ADELTA = FT & FT_ADELTA, TRACK = FT & FT_TRACK;
  using V = typename Vec<W>::T;
  using In = FeatIn<M, W, FT>;
  const int64_t B = A.B;
  const uint32_t flags = c.flags;
  const bool nt = (A.nt_stores & 1) != 0;
  constexpr int64_t TILE = (int64_t)BLOCK * W;
  const int64_t ntile = (B + TILE - 1) / TILE;
  int64_t it = blockIdx.x;
  if (it >= ntile) return;
  const bool ld_as = ADELTA && (flags & PCG_F_A_DELTA), ld_up = TRACK && (flags & PCG_F_REWARD_TRACK);
  auto load = [&](int64_t ee, In& q) {
#pragma unroll
    for (int i = 0; i < NX; ++i) q.x[i] = *reinterpret_cast<const V*>(A.x + (size_t)i * B + ee);
#pragma unroll
    for (int i = 0; i < NA; ++i) q.a[i] = *reinterpret_cast<const V*>(A.a + (size_t)i * B + ee);
    if constexpr (ADELTA) {
      if (ld_as) {
#pragma unroll
        for (int i = 0; i < NA; ++i) q.asave[i] = *reinterpret_cast<const V*>(A.a_save + (size_t)i * B + ee);
      }
    }
    if constexpr (TRACK) {
      if (ld_up) {
#pragma unroll
        for (int i = 0; i < NA; ++i) q.uprev[i] = *reinterpret_cast<const V*>(A.u_prev + (size_t)i * B + ee);
      }
    }
  };
  auto landall = [&](In& q) {
#pragma unroll
    for (int i = 0; i < NX; ++i) land(q.x[i]);
#pragma unroll
    for (int i = 0; i < NA; ++i) land(q.a[i]);
    if constexpr (ADELTA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) land(q.asave[i]);
    }
    if constexpr (TRACK) {
#pragma unroll
      for (int i = 0; i < NA; ++i) land(q.uprev[i]);
    }
  };
  In cur = {};
  int64_t e0 = it * TILE + (int64_t)threadIdx.x * W;
  bool live = e0 < B;
  if (live) load(e0, cur);
  for (;;) {
    landall(cur);
    const int64_t itn = it + gridDim.x;
    const int64_t e1 = itn * TILE + (int64_t)threadIdx.x * W;
    const bool live_n = (itn < ntile) && (e1 < B);
    In nxt = {};
    asm volatile("" ::: "memory");
    if (live_n) load(e1, nxt);
    asm volatile("" ::: "memory");
    if (live) {
      Pack<W> xs[NX], as[NA], sv[NA], up[NA];
#pragma unroll
      for (int j = 0; j < W; ++j) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[i].v[j] = Vec<W>::get(cur.x[i], j);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          as[i].v[j] = Vec<W>::get(cur.a[i], j);
          sv[i].v[j] = ADELTA ? Vec<W>::get(cur.asave[i], j) : 0.0;
          up[i].v[j] = TRACK ? Vec<W>::get(cur.uprev[i], j) : 0.0;
        }
      }
      FeatOut<M, W> out;
      env_step_feat<M, W, FT>(A, c, e0, A.t_scalar, as, sv, up, xs, out);
      store_feat<M, W, FT>(A, c, e0, xs, out, nt);
    }
    if (itn >= ntile) break;
    it = itn;
    e0 = e1;
    live = live_n;
    cur = nxt;
  }
}

// The masks built for every small model: the single features, constraints + tracking (the constraint-showcase
// configuration), the auto-reset launch of a lock-stepped episode, and "everything".  Mask 0 is the lean step with
// the `viol` output.
template <class M>
inline int feat_table(FeatEntry* out, int cap) {
  int n = 0;
  auto add = [&](unsigned m, StepFn f) {
    if (n < cap) out[n++] = FeatEntry{m, f};
  };
  add(0u, step_kernel_feat<M, 2, 0u>);
  add(FT_AR, step_kernel_feat<M, 2, FT_AR>);
  add(FT_CONS, step_kernel_feat<M, 2, FT_CONS>);
  add(FT_TRACK, step_kernel_feat<M, 2, FT_TRACK>);
  add(FT_CONS | FT_TRACK, step_kernel_feat<M, 2, FT_CONS | FT_TRACK>);
  add(FT_ALL, step_kernel_feat<M, 2, FT_ALL>);
  return n;
}

}  // namespace pcg
