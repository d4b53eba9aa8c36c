// pcg_kernels.hip -- fused environment-step kernels for gfx950 (MI355X) and the
// C ABI of include/pcgym_hip.h.
//
// Execution model
//   * one wavefront lane = one environment; 256-thread workgroups; grid = ceil(B/256)
//     (B = 2^20 -> 4096 workgroups = 16 per CU, 2 per CU per XCD round).
//   * all per-env arrays are SoA  field[component][B]  in HBM: lane i of a wave reads
//     address base + 8*i  -> every load/store instruction of a wave touches 512
//     contiguous bytes.  Each byte of state/action/obs crosses HBM exactly once per step.
//   * env state, the RK work vectors and the held inputs live in VGPRs; everything that
//     is identical for all envs of a plan (model constants, folded affine maps, bounds,
//     constraint rows) is one DevConst block in device memory read with wave-uniform
//     addresses, i.e. through the scalar cache into SGPRs -- no VGPR or LDS cost.
//   * time-indexed tables (set-point / disturbance schedules) are read with the scalar
//     unit when the batch is lock-stepped, and are staged in LDS when every env carries
//     its own step counter (per-lane table lookups after masked auto-reset).
//   * DOPRI5 stage vectors: VGPRs, or LDS [stage][component][lane] (PCG_OPT_LDS_STAGES).
//   * no MFMA: there is no dense contraction on this path.
//
// Reference path restated: make_env.step / reset (src/pcgym/pcgym.py:263-500),
// integration_engine (src/pcgym/integrator.py:65-107,163-182), model RHS
// (src/pcgym/model_classes.py) -- see pcg_models.hpp for per-model line ranges.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pcgym_hip.h"
#include "pcg_integrators.hpp"
#include "pcg_models.hpp"

// Optional per-wave timeline instrumentation (tools/timeline.py builds a second .so with
// -DPCG_TIMELINE): s_memtime stamps at kernel entry / loads landed / arithmetic done / stores issued
// / stores acknowledged, written for lane 0 of every wave into the `nsteps` debug buffer
// (8 x uint64 per wave).  Compiled out of the product library.
#ifdef PCG_TIMELINE
#define PCG_TL_DECL unsigned long long tl_[6] = {0, 0, 0, 0, 0, 0}
#define PCG_TL_STAMP(i) tl_[i] = __builtin_amdgcn_s_memtime()
#define PCG_TL_WAIT_STAMP(i)                      \
  do {                                            \
    __builtin_amdgcn_s_waitcnt(0);                \
    tl_[i] = __builtin_amdgcn_s_memtime();        \
  } while (0)
#define PCG_TL_FLUSH(e)                                                                   \
  do {                                                                                    \
    if ((threadIdx.x & 63) == 0 && A.nsteps) {                                            \
      unsigned long long* q = reinterpret_cast<unsigned long long*>(A.nsteps) + ((e) >> 6) * 8; \
      for (int i_ = 0; i_ < 5; ++i_) q[i_] = tl_[i_];                                     \
      q[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); /* HW_ID */     \
    }                                                                                     \
  } while (0)
#else
#define PCG_TL_DECL
#define PCG_TL_STAMP(i)
#define PCG_TL_WAIT_STAMP(i)
#define PCG_TL_FLUSH(e)
#endif

#ifndef PCG_LEAN_WPE
#define PCG_LEAN_WPE 1  // min waves per SIMD requested from the register allocator for the lean kernels
#endif

namespace pcg {

constexpr int BLOCK = 256;      // threads per workgroup (4 waves, one per SIMD)
constexpr int BLOCK_LDS = 64;   // LDS-staged DOPRI5: 6*NX*8 B of stage storage per lane
constexpr int tb(bool lds_stages) { return lds_stages ? BLOCK_LDS : BLOCK; }
// Minimum waves per SIMD asked of the register allocator.  DOPRI5 with <= 10 states needs ~280 registers
// when left alone (1 wave/SIMD, latency-bound: measured 14k cycles per attempted step against ~3.6k of
// issue); capping it at 256 costs a few spills and doubles the resident waves.
constexpr int wpe(int nx, int integ, bool lds_stages) {
  return (integ == PCG_INT_DOPRI5 && !lds_stages && nx <= 10) ? 2 : 1;
}
constexpr int KNU = PCG_MAX_NA + PCG_MAX_NDM;                      // kernel-side u width
constexpr int CON_W = PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + KNU; // padded constraint row

// Everything wave-uniform about a plan.  Read only through uniform addresses (scalar loads).
// Layout matters: the fields the lean step kernel touches are packed at the front and the affine
// maps are interleaved per row, so the compiler can fetch them with a few wide s_load_dwordx8/x16
// instead of ~50 separate 8-byte scalar loads (the scalar cache is shared by several CUs and every
// wave of the grid replays this prologue).
struct AMap {
  double pre, scale, off;  // act = (a + pre) * scale + off      (folds pcgym.py:372-379)
};
struct OMap {
  double lo, sc, off;      // obs = (o - lo) * sc + off          (pcgym.py:483-498; mask -> sc=off=0)
};
struct DevConst {
  double dt, h, rtol, atol;
  uint32_t flags;
  int32_t nx, na, ndm, nd, nsp, nsp_obs, ncon, nrew, N, substeps, max_steps, nobs, has_x0_unc, nunc;
  int32_t sp_index[PCG_MAX_NSP], d_slot[PCG_MAX_NDM];
  double kp[16];                      // model KP (pcg_models.hpp) of the five built-in models
  AMap amap[PCG_MAX_NA];
  double r_scale[PCG_MAX_NX];
  double d_default[PCG_MAX_NDM];
  OMap omap[PCG_MAX_NOBS];
  // ---- cold: a_delta, noise, reset, Gaussian disturbances, batch reward, constraints, affine model ----
  double a_act_lo[PCG_MAX_NA], a_act_hi[PCG_MAX_NA], a_0[PCG_MAX_NA];
  double noise_pct[PCG_MAX_NX];
  double x0[PCG_MAX_NX + PCG_MAX_NSP];
  double x0_unc[PCG_MAX_NX];
  double d_sigma[PCG_MAX_NDM], d_lo[PCG_MAX_NDM], d_hi[PCG_MAX_NDM];
  int32_t rew_index[PCG_MAX_NX];
  // constraint rows over [x(PCG_MAX_NX) | sp(PCG_MAX_NSP) | d(PCG_MAX_NDM) | u(KNU)], compat folded in
  double con_A[PCG_MAX_NCON][CON_W];
  double con_b[PCG_MAX_NCON];
  double kp_big[136];                 // KP of the affine custom model (A 8x8 | B 8x4 | c 8)
  // per-env parameter uncertainty (pcgym.py:212-253, 301-310): raw parameter vector + which entries vary
  double raw[32];
  double unc_pct[PCG_MAX_NUNC];
  int32_t unc_index[PCG_MAX_NUNC];
  int32_t emp_off[PCG_MAX_NUNC + 1];  // PCG_F_UNC_EMPIRICAL: sample tables live behind the schedules in `sched`
};

using CDevConst = const PCG_CONSTANT DevConst;

struct StepArgs {
  CDevConst* C;                       // constant address space: uniform reads -> s_load
  const PCG_CONSTANT double* sched;   // [nsp + nd][N]
  double* x;
  const double* a;
  const double* d;
  int32_t* t;
  double* a_save;
  double* obs;
  double* rew;
  uint8_t* done;
  uint8_t* viol;
  double* g;
  double* g_pre;
  int32_t* nsteps;
  double* p_unc;        // [nunc][B] per-env uncertain parameters
  const uint8_t* mask;  // reset only
  int64_t B;
  int64_t env_offset;
  uint64_t seed;
  int32_t t_scalar;
  int32_t sched_in_lds;  // per-env-t kernels: schedules staged in LDS
  int32_t nt_stores;     // stream kernels: non-temporal stores for obs / reward
  int32_t prio_mode;     // wave priority staggering (0 off)
  // rollout
  const double* a_seq;
  double* obs_seq;
  double* rew_seq;
  // element strides of the sequences: (step, component); the env index is always unit-stride
  int64_t a_ss, a_cs, o_ss, o_cs, r_ss;
  int32_t T;
};

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  RNG contract:
//   key = (seed_lo, seed_hi); ctr = (env_lo, env_hi, t, purpose + pair index)
//   one block -> two 53-bit uniforms -> one Box-Muller pair.
// ---------------------------------------------------------------------------
constexpr uint32_t RNG_NOISE = 0x100u, RNG_DIST = 0x200u, RNG_RESET = 0x300u;

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

PCG_DEV void rng_uniform2(uint64_t seed, uint64_t env, uint32_t t, uint32_t stream, double& u0, double& u1) {
  uint32_t o[4];
  philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), t, stream, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  u0 = (double)(((uint64_t)(o[0] >> 5) << 26) | (uint64_t)(o[1] >> 6)) * (1.0 / 9007199254740992.0);
  u1 = (double)(((uint64_t)(o[2] >> 5) << 26) | (uint64_t)(o[3] >> 6)) * (1.0 / 9007199254740992.0);
}

PCG_DEV void rng_normal2(uint64_t seed, uint64_t env, uint32_t t, uint32_t stream, double& z0, double& z1) {
  double u0, u1;
  rng_uniform2(seed, env, t, stream, u0, u1);
  const double r = sqrt(-2.0 * log(1.0 - u0));
  double s, c;
  sincospi(2.0 * u1, &s, &c);  // angle = 2*pi*u1, u1 in [0,1): no large-argument reduction
  z0 = r * c;
  z1 = r * s;
}

// Stagger the waves that share a SIMD: identical waves started together otherwise advance in
// lock-step (all load, all integrate, all store) and the memory system idles while the VALU works.
// Giving them distinct static priorities makes them finish one after the other, so stores and the
// next workgroups' loads overlap the remaining waves' arithmetic.
PCG_DEV void stagger_priority(int mode) {
  if (mode == 0) return;
  const unsigned k = (mode == 1) ? (blockIdx.x & 3u) : ((blockIdx.x >> 1) & 3u);
  switch (k) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// value at runtime index `idx` of a register array, without dynamic register indexing
template <int N>
PCG_DEV double pick(const double (&v)[N], int idx) {
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) r = (i == idx) ? v[i] : r;
  return r;
}

template <class M, class R = double, class K = typename M::CKP>
struct RhsFn {
  const K& kp;
  const typename M::template HoldT<R>& hold;
  PCG_DEV void operator()(const R (&x)[M::NX], R (&dx)[M::NX]) const { M::rhs(kp, hold, x, dx); }
};

// constraint rows g = A.[x|sp|d|u] - b  (affine form of the reference's callable, pcgym.py:560-577);
// writes rows to gout (if non-null) and returns "any row > 0".
template <class M>
PCG_DEV bool constraint_rows(CDevConst& c, const double (&x)[M::NX], const double (&spv)[PCG_MAX_NSP],
                             const double (&dv)[PCG_MAX_NDM], const double (&u)[M::NA + M::NDM], double* gout,
                             int64_t B, int64_t e) {
  bool violated = false;
  for (int r = 0; r < c.ncon; ++r) {
    const PCG_CONSTANT double* row = c.con_A[r];
    double g = -c.con_b[r];
#pragma unroll
    for (int i = 0; i < M::NX; ++i) g += row[i] * x[i];
#pragma unroll
    for (int k = 0; k < PCG_MAX_NSP; ++k) g += row[PCG_MAX_NX + k] * spv[k];
#pragma unroll
    for (int k = 0; k < PCG_MAX_NDM; ++k) g += row[PCG_MAX_NX + PCG_MAX_NSP + k] * dv[k];
#pragma unroll
    for (int j = 0; j < M::NA; ++j) g += row[PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + j] * u[j];
#pragma unroll
    for (int j = 0; j < M::NDM; ++j)
      g += row[PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + PCG_MAX_NA + j] * u[M::NA + j];
    if (gout) gout[(size_t)r * B + e] = g;
    violated |= (g > 0.0);
  }
  return violated;
}

// schedule lookup: lock-stepped -> uniform scalar load; per-env t -> LDS table (or global)
template <bool PER_ENV_T>
PCG_DEV double sched_at(const PCG_CONSTANT double* sched_g, const double* sched_l, bool in_lds, int row, int N, int idx) {
  if (PER_ENV_T && in_lds) return sched_l[row * N + idx];
  return sched_g[(size_t)row * N + idx];
}

// per-env results of one step, kept in registers so the caller chooses the store width
template <class M>
struct EnvOut {
  double ox[M::NX];          // observation rows of the physical states
  double osp[PCG_MAX_NSP];   // SP slots
  double od[PCG_MAX_NDM];    // disturbance slots
  double ounc[PCG_MAX_NUNC]; // uncertain-parameter slots
  double rew;
  bool done, viol;
};

// ---------------------------------------------------------------------------
// One env step for the lane's environment.  Statement order follows
// make_env.step (pcgym.py:350-500).  `x` is the lane's physical state (in/out);
// everything the hot path writes comes back in `out` (registers).  Only the rare
// side outputs (a_save, constraint rows, DOPRI5 step counts) are stored here.
// ---------------------------------------------------------------------------
// integrate one env over [0,dt] with model constants `kp` of any storage class (scalar constants of the
// plan, or a per-lane struct when parameters are uncertain)
template <class M, int INTEG, bool LDS_STAGES, class K>
PCG_DEV void integrate_env(const StepArgs& A, CDevConst& c, const K& kp, const double (&u)[M::NA + M::NDM],
                           double (&x)[M::NX], double* stage_l, int64_t e, int nx) {
  constexpr int NX = M::NX;
  const typename M::Hold hold = M::hold(kp, u);
  const RhsFn<M, double, K> f{kp, hold};
  if (INTEG == PCG_INT_RK4) {
    rk4<NX>(f, x, c.h, c.substeps);
  } else {
    int nacc = 0, nrej = 0;
    if (LDS_STAGES) {
      LdsStages<NX, BLOCK_LDS> Kst{stage_l + threadIdx.x};
      dopri5<NX>(f, Kst, x, nx, c.dt, c.rtol, c.atol, c.max_steps, nacc, nrej);
    } else {
      RegStages<NX> Kst;
      dopri5<NX>(f, Kst, x, nx, c.dt, c.rtol, c.atol, c.max_steps, nacc, nrej);
    }
    if (A.nsteps) {
      A.nsteps[e] = nacc;
      A.nsteps[A.B + e] = nrej;
    }
  }
}

template <class M, int INTEG, bool PER_ENV_T, bool LDS_STAGES, bool EXTRAS, bool UNC = false>
PCG_DEV void env_step(const StepArgs& A, CDevConst& c, const double* sched_l, double* stage_l, int64_t e,
                      int t, const double (&a_in)[M::NA], double (&x)[M::NX], EnvOut<M>& out) {
  constexpr int NX = M::NX, NA = M::NA, NDM = M::NDM;
  const int64_t B = A.B;
  const uint32_t flags = c.flags;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int N = c.N, nsp = c.nsp, nso = c.nsp_obs, nd = c.nd;
  const int tn = min(t + 1, N - 1);  // schedule index clamp (the reference would IndexError)
  const int tc = min(t, N - 1);
  const uint64_t env_id = (uint64_t)(A.env_offset + e);
  typename M::CKP& kp = *(typename M::CKP*)(M::DYNAMIC ? c.kp_big : c.kp);

  // ---- action map (pcgym.py:371-383) ----
  double u[NA + NDM];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    double av = 0.0;
    if (i < na) {
      av = (a_in[i] + c.amap[i].pre) * c.amap[i].scale + c.amap[i].off;
      if (EXTRAS && (flags & PCG_F_A_DELTA)) {
        av = A.a_save[(size_t)i * B + e] + av;  // Q2: the unclipped sum drives the plant
        A.a_save[(size_t)i * B + e] = fmin(fmax(av, c.a_act_lo[i]), c.a_act_hi[i]);
      }
    }
    u[i] = av;
  }
  // ---- disturbance injection (pcgym.py:386-412) ----
  double dv[PCG_MAX_NDM] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < NDM; ++j) u[NA + j] = c.d_default[j];
  if (NDM > 0 && nd > 0) {
#pragma unroll
    for (int k = 0; k < (NDM > 0 ? NDM : 1); ++k) {
      if (k < nd) {
        double v = (EXTRAS && A.d) ? A.d[(size_t)k * B + e] : sched_at<PER_ENV_T>(A.sched, sched_l, A.sched_in_lds, nsp + k, N, tn);
        if (EXTRAS && (flags & PCG_F_GAUSS_DIST)) {
          double z0, z1;
          rng_normal2(A.seed, env_id, (uint32_t)t, RNG_DIST + (uint32_t)(k >> 1), z0, z1);
          v += c.d_sigma[k] * ((k & 1) ? z1 : z0);
          v = fmin(fmax(v, c.d_lo[k]), c.d_hi[k]);
        }
        dv[k] = v;
        const int slot = c.d_slot[k];
#pragma unroll
        for (int j = 0; j < NDM; ++j) u[NA + j] = (j == slot) ? v : u[NA + j];
      }
    }
  }
  // ---- pre-step constraint check at t == 0 (pcgym.py:414-420) ----
  bool done = false;
  if (EXTRAS && c.ncon > 0 && t == 0) {
    double sp0[PCG_MAX_NSP];
#pragma unroll
    for (int k = 0; k < PCG_MAX_NSP; ++k) sp0[k] = (k < nso) ? c.x0[(M::DYNAMIC ? nx : NX) + k] : 0.0;
    const bool v = constraint_rows<M>(c, x, sp0, dv, u, A.g_pre, B, e);
    done = v && (flags & PCG_F_DONE_ON_CONS);
  }
  // ---- integrate over [0, dt], u held (pcgym.py:423-429, integrator.py:90-107,163-182) ----
  if constexpr (UNC && !M::DYNAMIC) {
    // per-env uncertain parameters (sampled at reset, pcgym.py:301-310): rebuild the folded model constants
    // for this lane from the raw parameter vector with the env's values substituted
    constexpr int NR = M::NRAW;
    double raw[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) raw[i] = c.raw[i];
    for (int j = 0; j < c.nunc; ++j) {
      const double v = A.p_unc[(size_t)j * B + e];
      out.ounc[j] = v;
      const int idx = c.unc_index[j];
#pragma unroll
      for (int i = 0; i < NR; ++i) raw[i] = (i == idx) ? v : raw[i];
    }
    typename M::KP kpl;
    double dd[PCG_MAX_NDM] = {0.0, 0.0, 0.0, 0.0};
    M::prep(raw, NX, NA, reinterpret_cast<double*>(&kpl), dd);
    if (c.ndm == 0) {  // unconfigured disturbance inputs take the (possibly uncertain) model parameters
#pragma unroll
      for (int j = 0; j < NDM; ++j) u[NA + j] = dd[j];
    }
    integrate_env<M, INTEG, LDS_STAGES>(A, c, kpl, u, x, stage_l, e, nx);
  } else {
    integrate_env<M, INTEG, LDS_STAGES>(A, c, kp, u, x, stage_l, e, nx);
  }
  // ---- SP slot uses SP[t_old] (pcgym.py:432-438, quirk Q5); t += 1 ----
  double spv[PCG_MAX_NSP] = {0.0, 0.0, 0.0, 0.0};
  double spn[PCG_MAX_NSP] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nsp) {
      spv[k] = sched_at<PER_ENV_T>(A.sched, sched_l, A.sched_in_lds, k, N, tc);
      spn[k] = sched_at<PER_ENV_T>(A.sched, sched_l, A.sched_in_lds, k, N, tn);
    }
  const int t_new = t + 1;
  // ---- post-step constraints (pcgym.py:443-446) ----
  bool violated = false;
  if (EXTRAS && c.ncon > 0) {
    violated = constraint_rows<M>(c, x, spv, dv, u, A.g, B, e);
    done |= violated && (flags & PCG_F_DONE_ON_CONS);
  }
  done |= (t_new == N - 1);  // pcgym.py:448-449
  out.done = done;
  out.viol = violated;
  // ---- reward on the noise-free state (pcgym.py:470-482) ----
  double r = 0.0;
  if (EXTRAS && (flags & PCG_F_REWARD_BATCH)) {  // pcgym.py:502-532
    if (t_new == N - 1) {
      for (int k = 0; k < c.nrew; ++k) {
        const double v = pick<NX>(x, c.rew_index[k]) * c.r_scale[k];
        r = (flags & PCG_F_MAXIMISE) ? r + v : r - v;
      }
      if ((flags & PCG_F_R_PENALTY) && violated) r -= 1000.0;
    }
  } else {  // pcgym.py:535-558
#pragma unroll
    for (int k = 0; k < PCG_MAX_NSP; ++k)
      if (k < nsp) {
        const double dd = pick<NX>(x, c.sp_index[k]) - spn[k];
        r += (-(dd * dd)) * c.r_scale[k];
        if ((flags & PCG_F_R_PENALTY) && violated) r -= 1000.0;  // Q4: once per SP key
      }
  }
  out.rew = r;
  // ---- observation: noise (pcgym.py:452-466), normalise (:483-489), mask (:495-498) ----
  double zn[NX];
  if (EXTRAS && (flags & PCG_F_NOISE)) {
#pragma unroll
    for (int i = 0; i < NX; i += 2) {
      double z0, z1;
      rng_normal2(A.seed, env_id, (uint32_t)t, RNG_NOISE + (uint32_t)(i >> 1), z0, z1);
      zn[i] = z0;
      if (i + 1 < NX) zn[i + 1] = z1;
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) {
      double o = x[i];
      if (EXTRAS && (flags & PCG_F_NOISE)) o += zn[i] * x[i] * c.noise_pct[i];
      out.ox[i] = (o - c.omap[i].lo) * c.omap[i].sc + c.omap[i].off;
    }
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nso) out.osp[k] = (spv[k] - c.omap[nx + k].lo) * c.omap[nx + k].sc + c.omap[nx + k].off;
#pragma unroll
  for (int k = 0; k < PCG_MAX_NDM; ++k)
    if (k < nd) out.od[k] = (dv[k] - c.omap[nx + nso + k].lo) * c.omap[nx + nso + k].sc + c.omap[nx + nso + k].off;
  if constexpr (UNC) {
    for (int j = 0; j < c.nunc; ++j) {
      const int q = nx + nso + nd + j;
      out.ounc[j] = (out.ounc[j] - c.omap[q].lo) * c.omap[q].sc + c.omap[q].off;
    }
  }
}

// scalar (8 B per lane) store of one env's outputs; obs_base = &obs[0][e] of the destination
template <class M, bool UNC = false>
PCG_DEV void store_obs(const StepArgs& A, CDevConst& c, const EnvOut<M>& out, double* obs_base, int64_t B) {
  // B here is the component stride of the destination
  const int nx = M::DYNAMIC ? c.nx : M::NX;
  const int nso = c.nsp_obs, nd = c.nd;
#pragma unroll
  for (int i = 0; i < M::NX; ++i)
    if (i < nx) obs_base[(size_t)i * B] = out.ox[i];
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nso) obs_base[(size_t)(nx + k) * B] = out.osp[k];
#pragma unroll
  for (int k = 0; k < PCG_MAX_NDM; ++k)
    if (k < nd) obs_base[(size_t)(nx + nso + k) * B] = out.od[k];
  if constexpr (UNC)
    for (int j = 0; j < c.nunc; ++j) obs_base[(size_t)(nx + nso + nd + j) * B] = out.ounc[j];
}

template <class M, bool UNC = false>
PCG_DEV void store_out(const StepArgs& A, CDevConst& c, int64_t e, const EnvOut<M>& out, double* obs_base) {
  const int64_t B = A.B;
  const int nx = M::DYNAMIC ? c.nx : M::NX;
  const int nso = c.nsp_obs, nd = c.nd;
#pragma unroll
  for (int i = 0; i < M::NX; ++i)
    if (i < nx) obs_base[(size_t)i * B] = out.ox[i];
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nso) obs_base[(size_t)(nx + k) * B] = out.osp[k];
#pragma unroll
  for (int k = 0; k < PCG_MAX_NDM; ++k)
    if (k < nd) obs_base[(size_t)(nx + nso + k) * B] = out.od[k];
  if constexpr (UNC)
    for (int j = 0; j < c.nunc; ++j) obs_base[(size_t)(nx + nso + nd + j) * B] = out.ounc[j];
  A.rew[e] = out.rew;
  A.done[e] = out.done ? 1 : 0;
  if (A.viol) A.viol[e] = out.viol ? 1 : 0;
}

// cooperative copy of the schedules into LDS (per-env-t kernels)
PCG_DEV void stage_schedules(const StepArgs& A, CDevConst& c, double* sched_l) {
  if (A.sched_in_lds) {
    const int n = (c.nsp + c.nd) * c.N;
    for (int i = threadIdx.x; i < n; i += blockDim.x) sched_l[i] = A.sched[i];
    __syncthreads();
  }
}

template <class M, int INTEG, bool PER_ENV_T, bool LDS_STAGES, bool EXTRAS, bool UNC = false>
__global__ __launch_bounds__(tb(LDS_STAGES), wpe(M::NX, INTEG, LDS_STAGES)) void step_kernel(const StepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  double* stage_l = lds;
  double* sched_l = lds + (LDS_STAGES ? 6 * NX * BLOCK_LDS : 0);
  if (PER_ENV_T) stage_schedules(A, c, sched_l);
  const int64_t e = (int64_t)blockIdx.x * tb(LDS_STAGES) + threadIdx.x;
  if (e >= A.B) return;
  const int64_t B = A.B;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int t = PER_ENV_T ? A.t[e] : A.t_scalar;
  stagger_priority(A.prio_mode);
  PCG_TL_DECL;
  PCG_TL_STAMP(0);
  double x[NX], a[NA];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = (i < nx) ? A.x[(size_t)i * B + e] : 0.0;
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = (i < na) ? A.a[(size_t)i * B + e] : 0.0;
  PCG_TL_WAIT_STAMP(1);  // loads landed
  EnvOut<M> out;
  env_step<M, INTEG, PER_ENV_T, LDS_STAGES, EXTRAS, UNC>(A, c, sched_l, stage_l, e, t, a, x, out);
  PCG_TL_STAMP(2);  // integration + epilogue arithmetic done
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) A.x[(size_t)i * B + e] = x[i];
  store_out<M, UNC>(A, c, e, out, A.obs + e);
  if (PER_ENV_T) A.t[e] = t + 1;
  PCG_TL_STAMP(3);       // stores issued
  PCG_TL_WAIT_STAMP(4);  // stores acknowledged
  PCG_TL_FLUSH(e);
}

// ---------------------------------------------------------------------------
// Streaming variant for the lean, lock-stepped hot path (BASELINE configs[1]):
//   * persistent grid (all workgroups resident), grid-stride over tiles of 256*EPL envs;
//   * EPL = 2 environments per lane -> every global access is 16 B per lane (dwordx4),
//     1 KiB contiguous per wave-instruction, and the two envs give the VALU two
//     independent dependency chains through exp/div;
//   * the loads of tile i+1 are issued before tile i is integrated, so each wave has HBM
//     reads in flight while it computes, and waves drift out of phase instead of
//     alternating chip-wide "all load / all compute / all store" rounds.
// Preconditions (checked on the host): no per-env t, no extras, no a_delta, no per-env d,
// B % EPL == 0 and 16-byte aligned rows when EPL == 2.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Lean env step for W envs per lane (Pack<W>): the hot path of BASELINE configs[1].
// Same statements as env_step with everything the lean plan cannot contain removed (a_delta,
// per-env / Gaussian disturbances, noise, constraints, terminal reward); the lock-stepped batch makes
// the SP / disturbance slots, `done` and all schedule values wave-uniform scalars.
// ---------------------------------------------------------------------------
template <int NX, int W>
PCG_DEV Pack<W> pick(const Pack<W> (&v)[NX], int idx) {
  Pack<W> r(0.0);
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j) r.v[j] = (i == idx) ? v[i].v[j] : r.v[j];
  return r;
}

template <class M, int W>
struct LeanOut {
  Pack<W> ox[M::NX];
  Pack<W> rew;
  double osp[PCG_MAX_NSP];  // wave-uniform
  double od[PCG_MAX_NDM];   // wave-uniform
  bool done;                // wave-uniform
};

template <class M, int W>
PCG_DEV void env_step_lean(const StepArgs& A, CDevConst& c, int t, const Pack<W> (&a_in)[M::NA],
                           Pack<W> (&x)[M::NX], LeanOut<M, W>& out) {
  constexpr int NX = M::NX, NA = M::NA, NDM = M::NDM;
  using R = Pack<W>;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int N = c.N, nsp = c.nsp, nso = c.nsp_obs, nd = c.nd;
  const int tn = min(t + 1, N - 1), tc = min(t, N - 1);
  typename M::CKP& kp = *(typename M::CKP*)(M::DYNAMIC ? c.kp_big : c.kp);
  // action map (pcgym.py:371-375) and held disturbance inputs (pcgym.py:386-404)
  R u[NA + NDM];
#pragma unroll
  for (int i = 0; i < NA; ++i)
    u[i] = (i < na) ? (a_in[i] + c.amap[i].pre) * c.amap[i].scale + c.amap[i].off : R(0.0);
  double ud[NDM > 0 ? NDM : 1];
#pragma unroll
  for (int j = 0; j < NDM; ++j) ud[j] = c.d_default[j];
#pragma unroll
  for (int k = 0; k < NDM; ++k)
    if (k < nd) {
      const double v = A.sched[(size_t)(nsp + k) * N + tn];  // Q6: index t+1
      out.od[k] = (v - c.omap[nx + nso + k].lo) * c.omap[nx + nso + k].sc + c.omap[nx + nso + k].off;
      const int slot = c.d_slot[k];
#pragma unroll
      for (int j = 0; j < NDM; ++j) ud[j] = (j == slot) ? v : ud[j];
    }
#pragma unroll
  for (int j = 0; j < NDM; ++j) u[NA + j] = R(ud[j]);
  // integrate over [0,dt] with the input held (integrator.py:163-182)
  const typename M::template HoldT<R> hold = M::template hold<R>(kp, u);
  const RhsFn<M, R> f{kp, hold};
  rk4<NX>(f, x, c.h, c.substeps);
  // SP slot = SP[t_old] (Q5), reward against SP[t_new] (pcgym.py:432-441, 535-558)
  R r(0.0);
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nsp) {
      const double spv = A.sched[(size_t)k * N + tc], spn = A.sched[(size_t)k * N + tn];
      if (k < nso) out.osp[k] = (spv - c.omap[nx + k].lo) * c.omap[nx + k].sc + c.omap[nx + k].off;
      const R dd = pick<NX, W>(x, c.sp_index[k]) - spn;
      r = r + (-(dd * dd)) * c.r_scale[k];
    }
  out.rew = r;
  out.done = (t + 1 == N - 1);  // pcgym.py:448-449
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) out.ox[i] = (x[i] - c.omap[i].lo) * c.omap[i].sc + c.omap[i].off;
}

template <int EPL>
struct Vec;
template <>
struct Vec<1> {
  using T = double;
  PCG_DEV static double get(const T& v, int) { return v; }
  PCG_DEV static T make(const double (&s)[1]) { return s[0]; }
  // streaming store: the data is not re-read by this kernel (obs / reward go to the policy)
  PCG_DEV static void store_nt(double* p, const double (&s)[1]) { __builtin_nontemporal_store(s[0], p); }
};
template <>
struct Vec<2> {
  using T = double2;
  typedef double d2 __attribute__((ext_vector_type(2)));
  PCG_DEV static double get(const T& v, int j) { return j ? v.y : v.x; }
  PCG_DEV static T make(const double (&s)[2]) { return make_double2(s[0], s[1]); }
  PCG_DEV static void store_nt(double* p, const double (&s)[2]) {
    __builtin_nontemporal_store(d2{s[0], s[1]}, reinterpret_cast<d2*>(p));
  }
};

PCG_DEV void land(double& v) { asm volatile("" : "+v"(v)); }
PCG_DEV void land(double2& v) {
  asm volatile("" : "+v"(v.x));
  asm volatile("" : "+v"(v.y));
}

template <class M, int INTEG, int EPL, int UNR>
__global__ __launch_bounds__(BLOCK, (PCG_LEAN_WPE > wpe(M::NX, INTEG, false) ? PCG_LEAN_WPE : wpe(M::NX, INTEG, false)))
void step_kernel_stream(const StepArgs A) {
  // One workgroup = UNR sub-tiles of 256*EPL envs.  All UNR sub-tiles' inputs are requested up front
  // (UNR * (NX+NA) loads in flight per lane), then the sub-tiles are integrated and stored one after
  // the other: the memory system works on sub-tile u+1.. while the VALU integrates sub-tile u, and
  // results leave as soon as each sub-tile is done instead of in one burst per wave.
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  using V = typename Vec<EPL>::T;
  const int64_t B = A.B;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int nso = c.nsp_obs;
  const int t = A.t_scalar;
  const bool nt = A.nt_stores != 0;
  stagger_priority(A.prio_mode);
  constexpr int64_t SUB = (int64_t)BLOCK * EPL;  // envs per sub-tile
  const int64_t tile = SUB * UNR;
  const int64_t ntile = (B + tile - 1) / tile;
  for (int64_t it = blockIdx.x; it < ntile; it += gridDim.x) {
    V xv[UNR][NX], av[UNR][NA];
    bool live[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t e0 = it * tile + u * SUB + (int64_t)threadIdx.x * EPL;
      live[u] = e0 < B;
      if (live[u]) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
          if (i < nx) xv[u][i] = *reinterpret_cast<const V*>(A.x + (size_t)i * B + e0);
#pragma unroll
        for (int i = 0; i < NA; ++i)
          if (i < na) av[u][i] = *reinterpret_cast<const V*>(A.a + (size_t)i * B + e0);
      }
    }
    // Land ALL inputs here, while only loads are outstanding.  gfx9-class hardware counts loads and
    // stores in one counter (vmcnt) and lets the two kinds complete out of order, so once a store is
    // pending the compiler can only wait with vmcnt(0) -- i.e. every later "wait for my input" would
    // also wait for the previous sub-tile's stores to be acknowledged (microseconds under load).
    // Passing the loaded registers through an empty asm makes this the single wait of the tile:
    // after it, the sub-tiles are integrated and stored back-to-back and no store is ever waited for.
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int i = 0; i < NX; ++i) land(xv[u][i]);
#pragma unroll
      for (int i = 0; i < NA; ++i) land(av[u][i]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t e0 = it * tile + u * SUB + (int64_t)threadIdx.x * EPL;
      if (!live[u]) continue;
      if constexpr (INTEG == PCG_INT_RK4) {
        // W = EPL envs advance together through one instruction stream (independent chains -> ILP)
        Pack<EPL> xs[NX], as[NA];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
          for (int j = 0; j < EPL; ++j) xs[i].v[j] = (i < nx) ? Vec<EPL>::get(xv[u][i], j) : 0.0;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < EPL; ++j) as[i].v[j] = (i < na) ? Vec<EPL>::get(av[u][i], j) : 0.0;
        LeanOut<M, EPL> out;
        env_step_lean<M, EPL>(A, c, t, as, xs, out);
        double tmp[EPL];
#pragma unroll
        for (int i = 0; i < NX; ++i)
          if (i < nx) {
            *reinterpret_cast<V*>(A.x + (size_t)i * B + e0) = Vec<EPL>::make(xs[i].v);
            if (nt) Vec<EPL>::store_nt(A.obs + (size_t)i * B + e0, out.ox[i].v);
            else *reinterpret_cast<V*>(A.obs + (size_t)i * B + e0) = Vec<EPL>::make(out.ox[i].v);
          }
#pragma unroll
        for (int k = 0; k < PCG_MAX_NSP; ++k)
          if (k < nso) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) tmp[j] = out.osp[k];
            if (nt) Vec<EPL>::store_nt(A.obs + (size_t)(nx + k) * B + e0, tmp);
            else *reinterpret_cast<V*>(A.obs + (size_t)(nx + k) * B + e0) = Vec<EPL>::make(tmp);
          }
#pragma unroll
        for (int k = 0; k < M::NDM; ++k)
          if (k < c.nd) {
#pragma unroll
            for (int j = 0; j < EPL; ++j) tmp[j] = out.od[k];
            if (nt) Vec<EPL>::store_nt(A.obs + (size_t)(nx + nso + k) * B + e0, tmp);
            else *reinterpret_cast<V*>(A.obs + (size_t)(nx + nso + k) * B + e0) = Vec<EPL>::make(tmp);
          }
        if (nt) Vec<EPL>::store_nt(A.rew + e0, out.rew.v);
        else *reinterpret_cast<V*>(A.rew + e0) = Vec<EPL>::make(out.rew.v);
        if (EPL == 2) *reinterpret_cast<uint16_t*>(A.done + e0) = out.done ? (uint16_t)0x0101u : (uint16_t)0;
        else A.done[e0] = out.done ? 1 : 0;
      } else {
      EnvOut<M> out[EPL];
      double xs[EPL][NX];
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        double a[NA];
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[j][i] = (i < nx) ? Vec<EPL>::get(xv[u][i], j) : 0.0;
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = (i < na) ? Vec<EPL>::get(av[u][i], j) : 0.0;
        env_step<M, INTEG, false, false, false>(A, c, nullptr, nullptr, e0 + j, t, a, xs[j], out[j]);
      }
      double tmp[EPL];
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (i < nx) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = xs[j][i];
          *reinterpret_cast<V*>(A.x + (size_t)i * B + e0) = Vec<EPL>::make(tmp);
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out[j].ox[i];
          if (nt) Vec<EPL>::store_nt(A.obs + (size_t)i * B + e0, tmp);
          else *reinterpret_cast<V*>(A.obs + (size_t)i * B + e0) = Vec<EPL>::make(tmp);
        }
#pragma unroll
      for (int k = 0; k < PCG_MAX_NSP; ++k)
        if (k < nso) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out[j].osp[k];
          if (nt) Vec<EPL>::store_nt(A.obs + (size_t)(nx + k) * B + e0, tmp);
          else *reinterpret_cast<V*>(A.obs + (size_t)(nx + k) * B + e0) = Vec<EPL>::make(tmp);
        }
#pragma unroll
      for (int k = 0; k < M::NDM; ++k)
        if (k < c.nd) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out[j].od[k];
          if (nt) Vec<EPL>::store_nt(A.obs + (size_t)(nx + nso + k) * B + e0, tmp);
          else *reinterpret_cast<V*>(A.obs + (size_t)(nx + nso + k) * B + e0) = Vec<EPL>::make(tmp);
        }
#pragma unroll
      for (int j = 0; j < EPL; ++j) tmp[j] = out[j].rew;
      if (nt) Vec<EPL>::store_nt(A.rew + e0, tmp);
      else *reinterpret_cast<V*>(A.rew + e0) = Vec<EPL>::make(tmp);
      if (EPL == 2) {
        *reinterpret_cast<uint16_t*>(A.done + e0) =
            (uint16_t)((out[0].done ? 1u : 0u) | (out[EPL - 1].done ? 0x100u : 0u));
      } else {
        A.done[e0] = out[0].done ? 1 : 0;
      }
      }  // INTEG
    }
  }
}

// ---------------------------------------------------------------------------
// Software-pipelined persistent variant of the lean kernel (PCG_OPT_VARIANT 4): each wave walks
// over its tiles and always has the NEXT tile's inputs in flight while it integrates the current one.
//   loop:  land(cur)            -- the only wait: cur's loads (issued one iteration ago) + previous stores
//          issue loads(next)
//          integrate(cur)       -- no memory operation inside the lean step
//          issue stores(cur)    -- never waited for explicitly
// The `land` placement matters: loads and stores share one in-order-per-kind counter (vmcnt), so the
// compiler can only wait with vmcnt(0) once stores are pending; waiting BEFORE the prefetch is issued
// keeps the prefetch out of that wait.
// ---------------------------------------------------------------------------
template <class M, int EPL>
__global__ __launch_bounds__(BLOCK, PCG_LEAN_WPE) void step_kernel_pipe(const StepArgs A) {
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  using V = typename Vec<EPL>::T;
  const int64_t B = A.B;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int nso = c.nsp_obs;
  const int t = A.t_scalar;
  const bool nt = A.nt_stores != 0;
  constexpr int64_t TILE = (int64_t)BLOCK * EPL;
  const int64_t ntile = (B + TILE - 1) / TILE;
  int64_t it = blockIdx.x;
  if (it >= ntile) return;
  V xv[NX], av[NA];
  int64_t e0 = it * TILE + (int64_t)threadIdx.x * EPL;
  bool live = e0 < B;
  auto load = [&](int64_t ee, V (&xd)[NX], V (&ad)[NA]) {
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (i < nx) xd[i] = *reinterpret_cast<const V*>(A.x + (size_t)i * B + ee);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (i < na) ad[i] = *reinterpret_cast<const V*>(A.a + (size_t)i * B + ee);
  };
  if (live) load(e0, xv, av);
  for (;;) {
#pragma unroll
    for (int i = 0; i < NX; ++i) land(xv[i]);
#pragma unroll
    for (int i = 0; i < NA; ++i) land(av[i]);
    const int64_t itn = it + gridDim.x;
    const int64_t e1 = itn * TILE + (int64_t)threadIdx.x * EPL;
    const bool live_n = (itn < ntile) && (e1 < B);
    V xn[NX], an[NA];
    asm volatile("" ::: "memory");
    if (live_n) load(e1, xn, an);
    asm volatile("" ::: "memory");
    if (live) {
      Pack<EPL> xs[NX], as[NA];
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < EPL; ++j) xs[i].v[j] = (i < nx) ? Vec<EPL>::get(xv[i], j) : 0.0;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < EPL; ++j) as[i].v[j] = (i < na) ? Vec<EPL>::get(av[i], j) : 0.0;
      LeanOut<M, EPL> out;
      env_step_lean<M, EPL>(A, c, t, as, xs, out);
      double tmp[EPL];
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (i < nx) {
          *reinterpret_cast<V*>(A.x + (size_t)i * B + e0) = Vec<EPL>::make(xs[i].v);
          if (nt) Vec<EPL>::store_nt(A.obs + (size_t)i * B + e0, out.ox[i].v);
          else *reinterpret_cast<V*>(A.obs + (size_t)i * B + e0) = Vec<EPL>::make(out.ox[i].v);
        }
#pragma unroll
      for (int k = 0; k < PCG_MAX_NSP; ++k)
        if (k < nso) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out.osp[k];
          if (nt) Vec<EPL>::store_nt(A.obs + (size_t)(nx + k) * B + e0, tmp);
          else *reinterpret_cast<V*>(A.obs + (size_t)(nx + k) * B + e0) = Vec<EPL>::make(tmp);
        }
#pragma unroll
      for (int k = 0; k < M::NDM; ++k)
        if (k < c.nd) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out.od[k];
          if (nt) Vec<EPL>::store_nt(A.obs + (size_t)(nx + nso + k) * B + e0, tmp);
          else *reinterpret_cast<V*>(A.obs + (size_t)(nx + nso + k) * B + e0) = Vec<EPL>::make(tmp);
        }
      if (nt) Vec<EPL>::store_nt(A.rew + e0, out.rew.v);
      else *reinterpret_cast<V*>(A.rew + e0) = Vec<EPL>::make(out.rew.v);
      if (EPL == 2) *reinterpret_cast<uint16_t*>(A.done + e0) = out.done ? (uint16_t)0x0101u : (uint16_t)0;
      else A.done[e0] = out.done ? 1 : 0;
    }
    if (itn >= ntile) break;
    it = itn;
    e0 = e1;
    live = live_n;
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = xn[i];
#pragma unroll
    for (int i = 0; i < NA; ++i) av[i] = an[i];
  }
}

// Lean fused rollout: T lock-stepped env steps, W envs per lane, state in registers throughout;
// per step only the action row(s) are read and reward (+ observation rows, if requested) written.
template <class M, int EPL>
__global__ __launch_bounds__(BLOCK) void rollout_kernel_lean(const StepArgs A) {
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  using V = typename Vec<EPL>::T;
  const int64_t B = A.B;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int nso = c.nsp_obs, nobs = c.nobs;
  const int64_t e0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * EPL;
  if (e0 >= B) return;
  Pack<EPL> xs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    V v;
    if (i < nx) v = *reinterpret_cast<const V*>(A.x + (size_t)i * B + e0);
#pragma unroll
    for (int j = 0; j < EPL; ++j) xs[i].v[j] = (i < nx) ? Vec<EPL>::get(v, j) : 0.0;
  }
  V an[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
    if (i < na) an[i] = *reinterpret_cast<const V*>(A.a_seq + (size_t)i * A.a_cs + e0);
  LeanOut<M, EPL> out;
  for (int s = 0; s < A.T; ++s) {
    Pack<EPL> as[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      land(an[i]);
#pragma unroll
      for (int j = 0; j < EPL; ++j) as[i].v[j] = (i < na) ? Vec<EPL>::get(an[i], j) : 0.0;
    }
    asm volatile("" ::: "memory");
    if (s + 1 < A.T) {  // next step's action in flight during this step's integration
      const double* nxt = A.a_seq + (size_t)(s + 1) * A.a_ss;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        if (i < na) an[i] = *reinterpret_cast<const V*>(nxt + (size_t)i * A.a_cs + e0);
    }
    asm volatile("" ::: "memory");
    env_step_lean<M, EPL>(A, c, A.t_scalar + s, as, xs, out);
    if (A.rew_seq) Vec<EPL>::store_nt(A.rew_seq + (size_t)s * A.r_ss + e0, out.rew.v);
    if (A.obs_seq) {
      double* o = A.obs_seq + (size_t)s * A.o_ss + e0;
      const int64_t ocs = A.o_cs;
      double tmp[EPL];
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (i < nx) Vec<EPL>::store_nt(o + (size_t)i * ocs, out.ox[i].v);
#pragma unroll
      for (int k = 0; k < PCG_MAX_NSP; ++k)
        if (k < nso) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out.osp[k];
          Vec<EPL>::store_nt(o + (size_t)(nx + k) * ocs, tmp);
        }
#pragma unroll
      for (int k = 0; k < M::NDM; ++k)
        if (k < c.nd) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) tmp[j] = out.od[k];
          Vec<EPL>::store_nt(o + (size_t)(nx + nso + k) * ocs, tmp);
        }
    }
  }
  // final state and the last step's outputs into the regular per-step buffers
  double tmp[EPL];
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) {
      *reinterpret_cast<V*>(A.x + (size_t)i * B + e0) = Vec<EPL>::make(xs[i].v);
      *reinterpret_cast<V*>(A.obs + (size_t)i * B + e0) = Vec<EPL>::make(out.ox[i].v);
    }
#pragma unroll
  for (int k = 0; k < PCG_MAX_NSP; ++k)
    if (k < nso) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) tmp[j] = out.osp[k];
      *reinterpret_cast<V*>(A.obs + (size_t)(nx + k) * B + e0) = Vec<EPL>::make(tmp);
    }
#pragma unroll
  for (int k = 0; k < M::NDM; ++k)
    if (k < c.nd) {
#pragma unroll
      for (int j = 0; j < EPL; ++j) tmp[j] = out.od[k];
      *reinterpret_cast<V*>(A.obs + (size_t)(nx + nso + k) * B + e0) = Vec<EPL>::make(tmp);
    }
  *reinterpret_cast<V*>(A.rew + e0) = Vec<EPL>::make(out.rew.v);
  if (EPL == 2) *reinterpret_cast<uint16_t*>(A.done + e0) = out.done ? (uint16_t)0x0101u : (uint16_t)0;
  else A.done[e0] = out.done ? 1 : 0;
}

// Open-loop fused rollout: T env steps with x in registers ("next" row f-1).
template <class M, int INTEG, bool LDS_STAGES>
__global__ __launch_bounds__(tb(LDS_STAGES)) void rollout_kernel(const StepArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  CDevConst& c = *A.C;
  constexpr int NX = M::NX, NA = M::NA;
  const int64_t e = (int64_t)blockIdx.x * tb(LDS_STAGES) + threadIdx.x;
  if (e >= A.B) return;
  const int64_t B = A.B;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  const int nobs = c.nobs;
  double x[NX], a[NA];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = (i < nx) ? A.x[(size_t)i * B + e] : 0.0;
  for (int s = 0; s < A.T; ++s) {
    const double* as = A.a_seq + (size_t)s * A.a_ss;
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = (i < na) ? as[(size_t)i * A.a_cs + e] : 0.0;
    const bool last = (s == A.T - 1);
    EnvOut<M> out;
    env_step<M, INTEG, false, LDS_STAGES, true>(A, c, lds, lds, e, A.t_scalar + s, a, x, out);
    if (A.rew_seq) A.rew_seq[(size_t)s * A.r_ss + e] = out.rew;
    if (A.obs_seq) store_obs<M>(A, c, out, A.obs_seq + (size_t)s * A.o_ss + e, A.o_cs);
    if (last || !A.obs_seq) store_out<M>(A, c, e, out, A.obs + e);  // io->obs/rew/done hold the last step
  }
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) A.x[(size_t)i * B + e] = x[i];
}

// reset (pcgym.py:263-349)
__global__ __launch_bounds__(BLOCK) void reset_kernel(const StepArgs A) {
  CDevConst& c = *A.C;
  const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= A.B) return;
  if (A.mask && !A.mask[e]) return;
  const int64_t B = A.B;
  const int nx = c.nx, nsp = c.nsp_obs, nd = c.nd;
  const uint64_t env_id = (uint64_t)(A.env_offset + e);
  for (int i = 0; i < nx; ++i) {
    double v = c.x0[i];
    if (c.has_x0_unc && c.x0_unc[i] != 0.0) {  // apply_uncertainties, pcgym.py:255-261
      const double pct = c.x0_unc[i];
      if (c.flags & PCG_F_X0_NORMAL) {
        double z0, z1;
        rng_normal2(A.seed, env_id, 0u, RNG_RESET + (uint32_t)(i >> 1), z0, z1);
        v = c.x0[i] + pct * c.x0[i] * ((i & 1) ? z1 : z0);
      } else {
        double u0, u1;
        rng_uniform2(A.seed, env_id, 0u, RNG_RESET + (uint32_t)(i >> 1), u0, u1);
        v = c.x0[i] * (1 + pct * (2.0 * ((i & 1) ? u1 : u0) - 1.0));
      }
    }
    A.x[(size_t)i * B + e] = v;
    A.obs[(size_t)i * B + e] = (v - c.omap[i].lo) * c.omap[i].sc + c.omap[i].off;
  }
  for (int k = 0; k < nsp; ++k)
    A.obs[(size_t)(nx + k) * B + e] = (c.x0[nx + k] - c.omap[nx + k].lo) * c.omap[nx + k].sc + c.omap[nx + k].off;
  for (int k = 0; k < nd; ++k) {  // disturbances[k][0] (pcgym.py:291-298, quirk Q6)
    const int j = nx + nsp + k;
    A.obs[(size_t)j * B + e] = (A.sched[(size_t)(c.nsp + k) * c.N] - c.omap[j].lo) * c.omap[j].sc + c.omap[j].off;
  }
  // uncertain model parameters (pcgym.py:301-310): sampled per env, appended to the observation
  for (int j = 0; j < c.nunc; ++j) {
    const double orig = c.raw[c.unc_index[j]], pct = c.unc_pct[j];
    const int ri = nx + j;  // RNG index after the x0 draws
    double v;
    if (c.flags & PCG_F_UNC_EMPIRICAL) {  // np.random.choice(samples), pcgym.py:311-316
      double u0, u1;
      rng_uniform2(A.seed, env_id, 0u, RNG_RESET + (uint32_t)(ri >> 1), u0, u1);
      const int len = c.emp_off[j + 1] - c.emp_off[j];
      int idx = (int)(((ri & 1) ? u1 : u0) * (double)len);
      idx = idx < len - 1 ? idx : len - 1;
      v = A.sched[(size_t)(c.nsp + c.nd) * c.N + c.emp_off[j] + idx];
    } else if (c.flags & PCG_F_X0_NORMAL) {
      double z0, z1;
      rng_normal2(A.seed, env_id, 0u, RNG_RESET + (uint32_t)(ri >> 1), z0, z1);
      v = orig + pct * orig * ((ri & 1) ? z1 : z0);
    } else {
      double u0, u1;
      rng_uniform2(A.seed, env_id, 0u, RNG_RESET + (uint32_t)(ri >> 1), u0, u1);
      v = orig * (1 + pct * (2.0 * ((ri & 1) ? u1 : u0) - 1.0));
    }
    A.p_unc[(size_t)j * B + e] = v;
    const int q = nx + nsp + nd + j;
    A.obs[(size_t)q * B + e] = (v - c.omap[q].lo) * c.omap[q].sc + c.omap[q].off;
  }
  if ((c.flags & PCG_F_A_DELTA) && A.a_save)
    for (int i = 0; i < c.na; ++i) A.a_save[(size_t)i * B + e] = c.a_0[i];
  if (A.t) A.t[e] = 0;
}

// test hooks ------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(BLOCK) void rhs_kernel(CDevConst* C, int64_t B, int nu_rows, const double* xg,
                                                    const double* ug, double* dxg) {
  CDevConst& c = *C;
  constexpr int NX = M::NX, NA = M::NA, NDM = M::NDM;
  const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= B) return;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  double x[NX], dx[NX], u[NA + NDM];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = (i < nx) ? xg[(size_t)i * B + e] : 0.0;
#pragma unroll
  for (int i = 0; i < NA; ++i) u[i] = (i < na) ? ug[(size_t)i * B + e] : 0.0;
#pragma unroll
  for (int j = 0; j < NDM; ++j) u[NA + j] = (na + j < nu_rows) ? ug[(size_t)(na + j) * B + e] : c.d_default[j];
  typename M::CKP& kp = *(typename M::CKP*)(M::DYNAMIC ? c.kp_big : c.kp);
  const typename M::Hold hold = M::hold(kp, u);
  M::rhs(kp, hold, x, dx);
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) dxg[(size_t)i * B + e] = dx[i];
}

template <class M, int INTEG, bool LDS_STAGES>
__global__ __launch_bounds__(tb(LDS_STAGES)) void integrate_kernel(CDevConst* C, int64_t B, int nu_rows,
                                                                   double* xg, const double* ug, int32_t* nsteps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  CDevConst& c = *C;
  constexpr int NX = M::NX, NA = M::NA, NDM = M::NDM;
  const int64_t e = (int64_t)blockIdx.x * tb(LDS_STAGES) + threadIdx.x;
  if (e >= B) return;
  const int nx = M::DYNAMIC ? c.nx : NX;
  const int na = M::DYNAMIC ? c.na : NA;
  double x[NX], u[NA + NDM];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = (i < nx) ? xg[(size_t)i * B + e] : 0.0;
#pragma unroll
  for (int i = 0; i < NA; ++i) u[i] = (i < na) ? ug[(size_t)i * B + e] : 0.0;
#pragma unroll
  for (int j = 0; j < NDM; ++j) u[NA + j] = (na + j < nu_rows) ? ug[(size_t)(na + j) * B + e] : c.d_default[j];
  typename M::CKP& kp = *(typename M::CKP*)(M::DYNAMIC ? c.kp_big : c.kp);
  const typename M::Hold hold = M::hold(kp, u);
  const RhsFn<M> f{kp, hold};
  if (INTEG == PCG_INT_RK4) {
    rk4<NX>(f, x, c.h, c.substeps);
  } else {
    int nacc = 0, nrej = 0;
    if (LDS_STAGES) {
      LdsStages<NX, BLOCK_LDS> K{lds + threadIdx.x};
      dopri5<NX>(f, K, x, nx, c.dt, c.rtol, c.atol, c.max_steps, nacc, nrej);
    } else {
      RegStages<NX> K;
      dopri5<NX>(f, K, x, nx, c.dt, c.rtol, c.atol, c.max_steps, nacc, nrej);
    }
    if (nsteps) {
      nsteps[e] = nacc;
      nsteps[B + e] = nrej;
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i)
    if (i < nx) xg[(size_t)i * B + e] = x[i];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
using StepFn = void (*)(const StepArgs);
using RhsKFn = void (*)(CDevConst*, int64_t, int, const double*, const double*, double*);
using IntKFn = void (*)(CDevConst*, int64_t, int, double*, const double*, int32_t*);

struct Kernels {
  StepFn step[PCG_INT_COUNT][2][2][2];  // [integrator][per_env_t][lds_stages][extras]
  StepFn stream[PCG_INT_COUNT][2][3];   // [integrator][EPL-1][log2 UNR]  (entries may be null)
  StepFn step_unc[PCG_INT_COUNT][2]; // per-env parameter uncertainty [integrator][per_env_t] (null for affine)
  StepFn pipe[2];                    // RK4 software-pipelined lean kernel [EPL-1] (may be null)
  StepFn roll_lean[2];               // RK4 lean fused rollout [EPL-1] (may be null)
  StepFn rollout[PCG_INT_COUNT][2];  // [integrator][lds_stages]
  RhsKFn rhs;
  IntKFn integ[PCG_INT_COUNT][2];
  int nx, na, ndm, nraw;
  bool dynamic, has_lds_stages;
  void (*prep)(const double*, int, int, double*, double*);
  size_t kp_bytes;
};

template <int ID>
Kernels make_kernels() {
  using M = Model<ID>;
  Kernels k;
  std::memset(&k, 0, sizeof(k));
  k.step[PCG_INT_RK4][0][0][0] = step_kernel<M, PCG_INT_RK4, false, false, false>;
  k.step[PCG_INT_RK4][0][0][1] = step_kernel<M, PCG_INT_RK4, false, false, true>;
  k.step[PCG_INT_RK4][1][0][0] = step_kernel<M, PCG_INT_RK4, true, false, false>;
  k.step[PCG_INT_RK4][1][0][1] = step_kernel<M, PCG_INT_RK4, true, false, true>;
  k.step[PCG_INT_DOPRI5][0][0][0] = step_kernel<M, PCG_INT_DOPRI5, false, false, false>;
  k.step[PCG_INT_DOPRI5][0][0][1] = step_kernel<M, PCG_INT_DOPRI5, false, false, true>;
  k.step[PCG_INT_DOPRI5][1][0][0] = step_kernel<M, PCG_INT_DOPRI5, true, false, false>;
  k.step[PCG_INT_DOPRI5][1][0][1] = step_kernel<M, PCG_INT_DOPRI5, true, false, true>;
  k.rollout[PCG_INT_RK4][0] = rollout_kernel<M, PCG_INT_RK4, false>;
  k.rollout[PCG_INT_DOPRI5][0] = rollout_kernel<M, PCG_INT_DOPRI5, false>;
  k.integ[PCG_INT_RK4][0] = integrate_kernel<M, PCG_INT_RK4, false>;
  k.integ[PCG_INT_DOPRI5][0] = integrate_kernel<M, PCG_INT_DOPRI5, false>;
  k.rhs = rhs_kernel<M>;
  if constexpr (!M::DYNAMIC) {
    k.step_unc[PCG_INT_RK4][0] = step_kernel<M, PCG_INT_RK4, false, false, true, true>;
    k.step_unc[PCG_INT_RK4][1] = step_kernel<M, PCG_INT_RK4, true, false, true, true>;
    k.step_unc[PCG_INT_DOPRI5][0] = step_kernel<M, PCG_INT_DOPRI5, false, false, true, true>;
    k.step_unc[PCG_INT_DOPRI5][1] = step_kernel<M, PCG_INT_DOPRI5, true, false, true, true>;
  }
  if constexpr (M::FULL) {
    // DOPRI5 with the stage vectors in LDS (PCG_OPT_LDS_STAGES)
    k.step[PCG_INT_DOPRI5][0][1][0] = step_kernel<M, PCG_INT_DOPRI5, false, true, false>;
    k.step[PCG_INT_DOPRI5][0][1][1] = step_kernel<M, PCG_INT_DOPRI5, false, true, true>;
    k.step[PCG_INT_DOPRI5][1][1][0] = step_kernel<M, PCG_INT_DOPRI5, true, true, false>;
    k.step[PCG_INT_DOPRI5][1][1][1] = step_kernel<M, PCG_INT_DOPRI5, true, true, true>;
    k.rollout[PCG_INT_DOPRI5][1] = rollout_kernel<M, PCG_INT_DOPRI5, true>;
    k.integ[PCG_INT_DOPRI5][1] = integrate_kernel<M, PCG_INT_DOPRI5, true>;
    // streaming / pipelined lean kernels
    k.roll_lean[0] = rollout_kernel_lean<M, 1>;
    k.stream[PCG_INT_RK4][0][0] = step_kernel_stream<M, PCG_INT_RK4, 1, 1>;
    k.stream[PCG_INT_DOPRI5][0][0] = step_kernel_stream<M, PCG_INT_DOPRI5, 1, 1>;
    // several envs per lane / sub-tiles per workgroup only where the per-env register footprint is
    // small (the HBM-bound models)
    if constexpr (M::NX <= 4) {
      k.roll_lean[1] = rollout_kernel_lean<M, 2>;
      k.pipe[0] = step_kernel_pipe<M, 1>;
      k.pipe[1] = step_kernel_pipe<M, 2>;
      k.stream[PCG_INT_RK4][0][1] = step_kernel_stream<M, PCG_INT_RK4, 1, 2>;
      k.stream[PCG_INT_RK4][0][2] = step_kernel_stream<M, PCG_INT_RK4, 1, 4>;
      k.stream[PCG_INT_RK4][1][0] = step_kernel_stream<M, PCG_INT_RK4, 2, 1>;
      k.stream[PCG_INT_RK4][1][1] = step_kernel_stream<M, PCG_INT_RK4, 2, 2>;
    }
  }
  // where no LDS-stage / RK4 variant exists the plain one is used
  for (int pe = 0; pe < 2; ++pe)
    for (int ex = 0; ex < 2; ++ex) {
      k.step[PCG_INT_RK4][pe][1][ex] = k.step[PCG_INT_RK4][pe][0][ex];
      if (!k.step[PCG_INT_DOPRI5][pe][1][ex]) k.step[PCG_INT_DOPRI5][pe][1][ex] = k.step[PCG_INT_DOPRI5][pe][0][ex];
    }
  k.rollout[PCG_INT_RK4][1] = k.rollout[PCG_INT_RK4][0];
  if (!k.rollout[PCG_INT_DOPRI5][1]) k.rollout[PCG_INT_DOPRI5][1] = k.rollout[PCG_INT_DOPRI5][0];
  k.integ[PCG_INT_RK4][1] = k.integ[PCG_INT_RK4][0];
  if (!k.integ[PCG_INT_DOPRI5][1]) k.integ[PCG_INT_DOPRI5][1] = k.integ[PCG_INT_DOPRI5][0];
  k.has_lds_stages = M::FULL;
  k.nx = M::NX;
  k.na = M::NA;
  k.ndm = M::NDM;
  k.nraw = M::NRAW;
  k.dynamic = M::DYNAMIC;
  k.prep = M::prep;
  k.kp_bytes = sizeof(typename M::KP);
  return k;
}

static const Kernels& kernels(int id) {
  static const Kernels K[PCG_MODEL_COUNT] = {
      make_kernels<PCG_MODEL_CSTR>(),        make_kernels<PCG_MODEL_FOUR_TANK>(),
      make_kernels<PCG_MODEL_ME>(),          make_kernels<PCG_MODEL_ME_REACTIVE>(),
      make_kernels<PCG_MODEL_CRYST>(),       make_kernels<PCG_MODEL_AFFINE>(),
      make_kernels<PCG_MODEL_COMPLEX_CSTR>(), make_kernels<PCG_MODEL_DISEASE>(),
      make_kernels<PCG_MODEL_BATCH>(),       make_kernels<PCG_MODEL_PHOTO>(),
      make_kernels<PCG_MODEL_CSTR_SERIES>(), make_kernels<PCG_MODEL_DISTILLATION>(),
      make_kernels<PCG_MODEL_POLYMER>(),     make_kernels<PCG_MODEL_BIOFILM>(),
      make_kernels<PCG_MODEL_HEAT_EX>(),     make_kernels<PCG_MODEL_INV_BATCH>(),
      make_kernels<PCG_MODEL_OSCILLATORS>()};
  return K[id];
}

// reference default parameters (model_classes.py:24-33, 877-889, 361-367, 777-786, 1260-1270)
static const double DEF_CSTR[] = {100, 100, 1000, 0.239, -5e4, 8750, 7.2e10, 5e4, 350, 1};
static const double DEF_FOUR_TANK[] = {9.81, 0.2, 0.2, 0.00085, 0.00095, 0.0035, 0.0030, 0.0020, 0.0025, 1, 1, 1, 1};
static const double DEF_ME[] = {5, 5, 1, 5, 2, 0.6, 0.05};
static const double DEF_ME_REACTIVE[] = {5.0, 5.0, 1.0, 0.01, 0.1, 2.0, 2.00, 0.00, 2.00, 0.00};
static const double DEF_CRYST[] = {0.923714966, -6754.878558, 0.92229965554, 1.341205945, 48.07514464, -4921.261419,
                                   1.871281405, 0.50523693,   7.271241375,   7.510905767, 2.658};
// model_classes.py:65-87, 156-158, 222-233, 443-453, 619-630, 689-695, 1172-1182
static const double DEF_COMPLEX_CSTR[] = {100, 100, 1000, 0.239, -5e4, 8750, 7.2e10, -3e4, 9000, 1.0e10, 5e4, 350, 1};
static const double DEF_DISEASE[] = {0.3, 0.1};
static const double DEF_BATCH[] = {1.0, 0.5, 5000, 6000, 8.314, -1000, -1500, 1000, 4.0, 100, 1.0};
static const double DEF_PHOTO[] = {0.0572, 0.0, 504.5, 0.00016, 0.281, 23.51, 16.89, 800.0, 178.9, 447.1, 393.1};
static const double DEF_CSTR_SERIES[] = {97.35, 298, 1e-3, 2e-3, 0.461, 0.732, 1.05e3, 3.766, 3.118e5, 46.14, 58.41, 8.3145e-3};
static const double DEF_DISTILLATION[] = {100.0, 1.0, 5.0, 0.2, 2000.0, 2000.0, 2000.0};
static const double DEF_POLYMER[] = {6e10, 4e10, 9e10, 7750, 8500, 8250, 0.5, 1.0, -3e4, 1200.0, 2.0};
// model_classes.py:1062-1073, 949-960, 269-272, 187-189
static const double DEF_BIOFILM[] = {10.0, 15.0, 1.5, 0.5, 1.0, 300, 0.8, 1.0, 0.5, 0.1, 1.5, 0.5};
static const double DEF_HEAT_EX[] = {1, 1, 1, 1, 2, 3, 1, 1, 1, 1, 1, 1};
static const double DEF_INV_BATCH[] = {55.0, 1.0, 2.0, 1.0};
static const double DEF_OSCILLATORS[] = {10, 1.0, 1.0};
static const double* const DEFAULTS[] = {DEF_CSTR,        DEF_FOUR_TANK, DEF_ME,    DEF_ME_REACTIVE, DEF_CRYST,
                                         nullptr,         DEF_COMPLEX_CSTR, DEF_DISEASE, DEF_BATCH, DEF_PHOTO,
                                         DEF_CSTR_SERIES, DEF_DISTILLATION, DEF_POLYMER, DEF_BIOFILM,
                                         DEF_HEAT_EX,     DEF_INV_BATCH,    DEF_OSCILLATORS};

}  // namespace pcg

using namespace pcg;

struct pcg_plan {
  uint32_t magic;
  int device;
  int model_id, integrator_id;
  int lds_stages;
  int variant;       // PCG_OPT_VARIANT: 0 auto, 1 classic, 2 stream EPL=1, 3 stream EPL=2
  int stream_bpc;    // PCG_OPT_STREAM_BLOCKS_PER_CU: 0 = occupancy query
  int nt_stores;     // PCG_OPT_NT_STORES
  int prio_mode;     // PCG_OPT_PRIO_STAGGER
  int num_cus;
  int stream_occ[2][3]; // resident workgroups per CU of the stream kernels (0 = not queried yet)
  int stream_unr;    // PCG_OPT_STREAM_UNROLL: log2(sub-tiles per workgroup)
  int pipe_occ[2];
  int64_t env_offset;
  DevConst hc;       // host copy
  DevConst* dC;      // device copy
  double* dsched;    // [nsp+nd][N]
  size_t sched_bytes;
  int cfg_nu;        // na + ndm as the caller counts them
};
static constexpr uint32_t PLAN_MAGIC = 0x50434731u;  // 'PCG1'

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return (int)_e;      \
  } while (0)

extern "C" {

int pcg_version(void) { return PCG_ABI_VERSION; }

const char* pcg_strerror(int status) {
  switch (status) {
    case PCG_OK: return "ok";
    case PCG_E_NULL: return "required pointer is NULL";
    case PCG_E_MODEL: return "unknown model or integrator id";
    case PCG_E_DIM: return "dimension out of range or inconsistent";
    case PCG_E_VALUE: return "invalid scalar value";
    case PCG_E_PLAN: return "invalid plan handle or wrong device";
    case PCG_E_UNSUPPORTED: return "combination not supported by this build";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "unknown status";
}

int pcg_model_info(int model_id, int32_t* nx, int32_t* nu, int32_t* ndm, int32_t* n_params) {
  if (model_id < 0 || model_id >= PCG_MODEL_COUNT) return PCG_E_MODEL;
  const Kernels& k = kernels(model_id);
  if (nx) *nx = k.nx;
  if (nu) *nu = k.na;
  if (ndm) *ndm = k.ndm;
  if (n_params) *n_params = k.nraw;
  return PCG_OK;
}

int pcg_model_default_params(int model_id, double* out, int32_t n_out) {
  if (model_id < 0 || model_id >= PCG_MODEL_COUNT) return PCG_E_MODEL;
  if (!out) return PCG_E_NULL;
  const Kernels& k = kernels(model_id);
  if (k.nraw < 0 || !DEFAULTS[model_id]) return PCG_E_UNSUPPORTED;
  if (n_out < k.nraw) return PCG_E_DIM;
  for (int i = 0; i < k.nraw; ++i) out[i] = DEFAULTS[model_id][i];
  return PCG_OK;
}

void pcg_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t o[4];
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], o);
  for (int i = 0; i < 4; ++i) out[i] = o[i];
}

// Validates cfg and fills the host DevConst.  No HIP calls: unit-testable without a GPU.
static int build_devconst(const pcg_env_cfg* c, DevConst* d, int* cfg_nu_out) {
  if (!c || !d) return PCG_E_NULL;
  if (c->model_id < 0 || c->model_id >= PCG_MODEL_COUNT) return PCG_E_MODEL;
  if (c->integrator_id < 0 || c->integrator_id >= PCG_INT_COUNT) return PCG_E_MODEL;
  const Kernels& k = kernels(c->model_id);
  const int nx = c->nx, na = c->na, ndm = c->ndm, nd = c->nd, nsp = c->nsp, ncon = c->ncon, nrew = c->nrew;
  if (k.dynamic) {
    if (nx < 1 || nx > k.nx || na < 1 || na > k.na || ndm != 0) return PCG_E_DIM;
    if (c->n_params != nx * nx + nx * na + nx) return PCG_E_DIM;
  } else {
    if (nx != k.nx || na != k.na) return PCG_E_DIM;
    if (ndm != 0 && ndm != k.ndm) return PCG_E_DIM;
    if (c->n_params != k.nraw) return PCG_E_DIM;
    // coupled_oscillators: the ring size is a structural parameter, only the reference default N = 10 is compiled
    if (c->model_id == PCG_MODEL_OSCILLATORS && (!c->params || c->params[0] != 10.0)) return PCG_E_UNSUPPORTED;
  }
  if (nd < 0 || nd > ndm || nsp < 0 || nsp > PCG_MAX_NSP || ncon < 0 || ncon > PCG_MAX_NCON) return PCG_E_DIM;
  const int nso = c->nsp_obs;
  const int nunc = c->nunc;
  if (nunc < 0 || nunc > PCG_MAX_NUNC) return PCG_E_DIM;
  if (nunc > 0 && (k.dynamic || !c->unc_index || !c->unc_pct)) return nunc > 0 && k.dynamic ? PCG_E_UNSUPPORTED : PCG_E_NULL;
  if (nunc > 0 && nd > 0) return PCG_E_UNSUPPORTED;  // quirk Q11: the reference's slot layout is inconsistent there
  if (nso != 0 && nso != nsp) return PCG_E_DIM;
  if (nd > 0 && nso != nsp) return PCG_E_UNSUPPORTED;
  if (nrew < 0 || nrew > PCG_MAX_NX) return PCG_E_DIM;
  if (c->N < 2 || c->N > PCG_MAX_N) return PCG_E_DIM;
  if (!(c->dt > 0.0) || !std::isfinite(c->dt)) return PCG_E_VALUE;
  if (c->integrator_id == PCG_INT_RK4 && c->substeps < 0) return PCG_E_VALUE;  // 0 = no integration (I/O probe)
  if (c->integrator_id == PCG_INT_DOPRI5 && (!(c->rtol > 0) || !(c->atol >= 0) || c->max_steps < 1))
    return PCG_E_VALUE;
  const int nobs = nx + nso + nd + nunc, cnu = na + ndm;
  if (!c->params || !c->x0 || !c->a_low || !c->a_high || !c->o_low || !c->o_high) return PCG_E_NULL;
  if (nsp && (!c->sp_index || !c->sp)) return PCG_E_NULL;
  if ((nsp || nrew) && !c->r_scale) return PCG_E_NULL;
  if (nrew && !c->rew_index) return PCG_E_NULL;
  if (nd && (!c->d_slot || !c->d_sched)) return PCG_E_NULL;
  if (ndm && !c->d_default) return PCG_E_NULL;
  if (ncon && (!c->con_A || !c->con_b)) return PCG_E_NULL;
  if ((c->flags & PCG_F_A_DELTA) && (!c->a_act_low || !c->a_act_high || !c->a_0)) return PCG_E_NULL;
  if ((c->flags & PCG_F_NOISE) && !c->noise_pct) return PCG_E_NULL;
  if ((c->flags & PCG_F_GAUSS_DIST) && nd && (!c->d_sigma || !c->d_clip_lo || !c->d_clip_hi)) return PCG_E_NULL;

  std::memset(d, 0, sizeof(*d));
  double ddef[PCG_MAX_NDM] = {0, 0, 0, 0};
  k.prep(c->params, nx, na, k.dynamic ? d->kp_big : d->kp, ddef);
  for (int j = 0; j < k.ndm; ++j) d->d_default[j] = ndm ? c->d_default[j] : ddef[j];
  const bool norm_a = c->flags & PCG_F_NORMALISE_A, norm_o = c->flags & PCG_F_NORMALISE_O;
  const bool compat = c->flags & PCG_F_REF_COMPAT;
  for (int i = 0; i < na; ++i) {
    const double lo = c->a_low[i], hs = (c->a_high[i] - c->a_low[i]) / 2;
    if (!norm_a) {
      d->amap[i] = AMap{0, 1, 0};
    } else if ((c->flags & PCG_F_A_DELTA) && compat) {
      // Q1 (pcgym.py:372-379): f(f(a)), f(a) = (a+1)*hs + lo
      d->amap[i] = AMap{1, hs * hs, (lo + 1) * hs + lo};
    } else {
      d->amap[i] = AMap{1, hs, lo};
    }
    if (c->flags & PCG_F_A_DELTA) {
      d->a_act_lo[i] = c->a_act_low[i]; d->a_act_hi[i] = c->a_act_high[i]; d->a_0[i] = c->a_0[i];
    }
  }
  for (int i = 0; i < nobs; ++i) {
    const bool masked = (i < nx) && c->obs_mask && !c->obs_mask[i];
    if (masked) {
      d->omap[i] = OMap{0, 0, 0};
    } else if (norm_o) {
      if (!(c->o_high[i] > c->o_low[i])) return PCG_E_VALUE;
      d->omap[i] = OMap{c->o_low[i], 2.0 / (c->o_high[i] - c->o_low[i]), -1.0};
    } else {
      d->omap[i] = OMap{0, 1, 0};
    }
  }
  for (int i = 0; i < nsp; ++i) {
    if (c->sp_index[i] < 0 || c->sp_index[i] >= nx) return PCG_E_DIM;
    d->sp_index[i] = c->sp_index[i];
  }
  for (int i = 0; i < nrew; ++i) {
    if (c->rew_index[i] < 0 || c->rew_index[i] >= nx) return PCG_E_DIM;
    d->rew_index[i] = c->rew_index[i];
  }
  const int nrs = (c->flags & PCG_F_REWARD_BATCH) ? nrew : nsp;
  for (int i = 0; i < nrs; ++i) d->r_scale[i] = c->r_scale[i];
  if (c->flags & PCG_F_NOISE)
    for (int i = 0; i < nx; ++i) d->noise_pct[i] = c->noise_pct[i];
  for (int i = 0; i < nx + nso; ++i) d->x0[i] = c->x0[i];
  d->has_x0_unc = c->x0_unc ? 1 : 0;
  if (c->x0_unc)
    for (int i = 0; i < nx; ++i) d->x0_unc[i] = c->x0_unc[i];
  for (int i = 0; i < nd; ++i) {
    if (c->d_slot[i] < 0 || c->d_slot[i] >= ndm) return PCG_E_DIM;
    d->d_slot[i] = c->d_slot[i];
    if (c->flags & PCG_F_GAUSS_DIST) {
      d->d_sigma[i] = c->d_sigma[i]; d->d_lo[i] = c->d_clip_lo[i]; d->d_hi[i] = c->d_clip_hi[i];
    }
  }
  // constraint rows: cfg layout [state(nobs) | uk(cnu)] -> padded kernel layout; compat Q3 folded:
  //   state' = (s+1)*hs + lo = s*hs + (hs+lo)   (pcgym.py:601-608), input' likewise with a_space (:597-600)
  for (int r = 0; r < ncon; ++r) {
    const double* row = c->con_A + (size_t)r * (nobs + cnu);
    double b = c->con_b[r];
    for (int i = 0; i < nobs; ++i) {
      double coef = row[i];
      if (compat && norm_o) {
        const double hs = (c->o_high[i] - c->o_low[i]) / 2;
        b -= coef * (hs + c->o_low[i]);
        coef *= hs;
      }
      if (i >= nx + nso + nd) {  // uncertain-parameter slots cannot enter constraint rows
        if (coef != 0.0) return PCG_E_UNSUPPORTED;
        continue;
      }
      const int col = (i < nx) ? i : (i < nx + nso) ? PCG_MAX_NX + (i - nx) : PCG_MAX_NX + PCG_MAX_NSP + (i - nx - nso);
      d->con_A[r][col] = coef;
    }
    for (int j = 0; j < cnu; ++j) {
      double coef = row[nobs + j];
      if (compat && norm_a) {
        if (cnu != na && na != 1) return PCG_E_UNSUPPORTED;  // the reference itself raises (broadcast error)
        const int q = (na == 1) ? 0 : j;
        const double hs = (c->a_high[q] - c->a_low[q]) / 2;
        b -= coef * (hs + c->a_low[q]);
        coef *= hs;
      }
      const int col = PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + ((j < na) ? j : PCG_MAX_NA + (j - na));
      d->con_A[r][col] = coef;
    }
    d->con_b[r] = b;
  }
  d->nunc = nunc;
  if (!k.dynamic)
    for (int i = 0; i < k.nraw && i < 32; ++i) d->raw[i] = c->params[i];
  for (int j = 0; j < nunc; ++j) {
    if (c->unc_index[j] < 0 || c->unc_index[j] >= k.nraw) return PCG_E_DIM;
    d->unc_index[j] = c->unc_index[j];
    d->unc_pct[j] = c->unc_pct[j];
    if (c->flags & PCG_F_UNC_EMPIRICAL) {
      if (!c->unc_emp || !c->unc_emp_off) return PCG_E_NULL;
      if (c->unc_emp_off[0] != 0 || c->unc_emp_off[j + 1] <= c->unc_emp_off[j] || c->unc_emp_off[j + 1] > PCG_MAX_EMP)
        return PCG_E_DIM;
      d->emp_off[j] = c->unc_emp_off[j];
      d->emp_off[j + 1] = c->unc_emp_off[j + 1];
    }
  }
  d->dt = c->dt;
  d->h = c->dt / (c->substeps > 0 ? c->substeps : 1);
  d->rtol = c->rtol;
  d->atol = c->atol;
  d->nx = nx; d->na = na; d->ndm = ndm; d->nd = nd; d->nsp = nsp; d->nsp_obs = nso; d->ncon = ncon; d->nrew = nrew;
  d->N = c->N; d->substeps = c->substeps; d->max_steps = c->max_steps; d->nobs = nobs;
  d->flags = c->flags;
  if (cfg_nu_out) *cfg_nu_out = cnu;
  return PCG_OK;
}

int pcg_plan_create(pcg_plan** out, const pcg_env_cfg* cfg) {
  if (!out || !cfg) return PCG_E_NULL;
  *out = nullptr;
  pcg_plan* p = new (std::nothrow) pcg_plan();
  if (!p) return (int)hipErrorOutOfMemory;
  int rc = build_devconst(cfg, &p->hc, &p->cfg_nu);
  if (rc != PCG_OK) {
    delete p;
    return rc;
  }
  p->magic = PLAN_MAGIC;
  p->model_id = cfg->model_id;
  p->integrator_id = cfg->integrator_id;
  p->lds_stages = 0;
  p->variant = 0;
  p->stream_bpc = 0;
  p->nt_stores = 1;  // measured: 20.3 -> 18.7 us per launch on the cstr workload (profiles/)
  p->prio_mode = 0;
  p->num_cus = 0;
  for (auto& r : p->stream_occ) for (int& v : r) v = 0;
  p->stream_unr = 0;
  p->pipe_occ[0] = p->pipe_occ[1] = 0;
  p->env_offset = 0;
  p->dC = nullptr;
  p->dsched = nullptr;
  hipError_t e = hipGetDevice(&p->device);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&p->num_cus, hipDeviceAttributeMultiprocessorCount, p->device);
  if (e != hipSuccess) { delete p; return (int)e; }
  const int rows = cfg->nsp + cfg->nd;
  const bool emp = (cfg->flags & PCG_F_UNC_EMPIRICAL) && cfg->nunc > 0;
  const size_t n_emp = emp ? (size_t)cfg->unc_emp_off[cfg->nunc] : 0;
  p->sched_bytes = sizeof(double) * (size_t)(rows > 0 ? rows : 1) * cfg->N;
  const size_t sched_alloc = sizeof(double) * ((size_t)(rows > 0 ? rows : 1) * cfg->N + n_emp);
  e = hipMalloc((void**)&p->dC, sizeof(DevConst));
  if (e == hipSuccess) e = hipMalloc((void**)&p->dsched, sched_alloc);
  if (e == hipSuccess) e = hipMemcpy(p->dC, &p->hc, sizeof(DevConst), hipMemcpyHostToDevice);
  if (e == hipSuccess && cfg->nsp)
    e = hipMemcpy(p->dsched, cfg->sp, sizeof(double) * (size_t)cfg->nsp * cfg->N, hipMemcpyHostToDevice);
  if (e == hipSuccess && cfg->nd)
    e = hipMemcpy(p->dsched + (size_t)cfg->nsp * cfg->N, cfg->d_sched, sizeof(double) * (size_t)cfg->nd * cfg->N,
                  hipMemcpyHostToDevice);
  if (e == hipSuccess && n_emp)  // empirical sample tables behind the schedule rows
    e = hipMemcpy(p->dsched + (size_t)rows * cfg->N, cfg->unc_emp, sizeof(double) * n_emp, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (p->dC) (void)hipFree(p->dC);
    if (p->dsched) (void)hipFree(p->dsched);
    delete p;
    return (int)e;
  }
  *out = p;
  return PCG_OK;
}

static bool plan_ok(const pcg_plan* p) { return p && p->magic == PLAN_MAGIC; }

int pcg_plan_destroy(pcg_plan* p) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  p->magic = 0;
  hipError_t e1 = hipFree(p->dC), e2 = hipFree(p->dsched);
  delete p;
  if (e1 != hipSuccess) return (int)e1;
  if (e2 != hipSuccess) return (int)e2;
  return PCG_OK;
}

int pcg_plan_set_env_offset(pcg_plan* p, int64_t env_offset) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  p->env_offset = env_offset;
  return PCG_OK;
}

int pcg_plan_set_option(pcg_plan* p, int option, int64_t value) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  switch (option) {
    case PCG_OPT_ENV_OFFSET: p->env_offset = value; return PCG_OK;
    case PCG_OPT_LDS_STAGES: p->lds_stages = value ? 1 : 0; return PCG_OK;
    case PCG_OPT_STREAM_BLOCKS_PER_CU: p->stream_bpc = (int)value; return PCG_OK;
    case PCG_OPT_NT_STORES: p->nt_stores = value ? 1 : 0; return PCG_OK;
    case PCG_OPT_STREAM_UNROLL: p->stream_unr = (int)value; return PCG_OK;
    case PCG_OPT_PRIO_STAGGER: p->prio_mode = (int)value; return PCG_OK;
    case PCG_OPT_VARIANT:
      if (value < 0 || value > 4) return PCG_E_VALUE;
      p->variant = (int)value;
      return PCG_OK;
    default: return PCG_E_VALUE;
  }
}

int64_t pcg_plan_bytes_per_env_step(const pcg_plan* p, const pcg_buffers* io) {
  if (!plan_ok(p) || !io) return PCG_E_PLAN;
  const DevConst& c = p->hc;
  // read x, read a, write x', write obs, write reward, done (+viol)
  int64_t A = 8 * (int64_t)(c.nx + c.na + c.nx + c.nobs + 1) + 1;
  if (io->viol) A += 1;
  if (io->d) A += 8 * c.nd;
  if (io->g) A += 8 * c.ncon;
  if (io->t) A += 8;
  if ((c.flags & PCG_F_A_DELTA) && io->a_save) A += 16 * c.na;
  if (io->nsteps) A += 8;
  A += 16 * c.nunc;  // per-env parameters read + their observation slots written
  return A;
}

static int fill_args(const pcg_plan* p, const pcg_buffers* io, StepArgs* a) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!io) return PCG_E_NULL;
  if (io->B < 0) return PCG_E_DIM;
  std::memset(a, 0, sizeof(*a));
  a->C = (CDevConst*)p->dC; a->sched = (const PCG_CONSTANT double*)p->dsched;
  a->x = io->x; a->a = io->a; a->d = io->d; a->t = io->t; a->a_save = io->a_save; a->obs = io->obs;
  a->rew = io->rew; a->done = io->done; a->viol = io->viol; a->g = io->g; a->g_pre = io->g_pre;
  a->nsteps = io->nsteps; a->B = io->B; a->env_offset = p->env_offset;
  a->p_unc = io->p_unc;
  a->prio_mode = p->prio_mode;
  return PCG_OK;
}

static inline unsigned grid_for(int64_t B, int block = BLOCK) { return (unsigned)((B + block - 1) / block); }

// Resident 256-thread workgroups per CU (= waves per SIMD) of a persistent kernel; < 0: -(hipError_t).
// The occupancy API over-reports by one for some register counts on ROCm 7.2 (MI355X_MICROARCH.md
// "Residency"), and a persistent grid with a non-resident workgroup serialises a whole extra round:
// bound it by the VGPR allocation too.
static int resident_blocks(StepFn fn) {
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, BLOCK, 0);
  if (e != hipSuccess) return -(int)e;
  hipFuncAttributes fa;
  e = hipFuncGetAttributes(&fa, (const void*)fn);
  if (e != hipSuccess) return -(int)e;
  const int alloc = ((fa.numRegs + 7) / 8) * 8;
  const int by_vgpr = alloc > 0 ? 512 / alloc : 8;
  if (nb > by_vgpr) nb = by_vgpr;
  if (nb > 8) nb = 8;
  return nb > 0 ? nb : 1;
}

// Fill every lazily queried occupancy of the plan's candidate persistent kernels (done before a stream
// capture so that no query runs while capturing).
static int warm_occupancy(pcg_plan* p) {
  const Kernels& k = kernels(p->model_id);
  for (int e = 0; e < 2; ++e) {
    if (p->integrator_id == PCG_INT_RK4 && k.pipe[e] && p->pipe_occ[e] == 0) {
      const int q = resident_blocks(k.pipe[e]);
      if (q < 0) return -q;
      p->pipe_occ[e] = q;
    }
    for (int lu = 0; lu < 3; ++lu)
      if (k.stream[p->integrator_id][e][lu] && p->stream_occ[e][lu] == 0) {
        const int q = resident_blocks(k.stream[p->integrator_id][e][lu]);
        if (q < 0) return -q;
        p->stream_occ[e][lu] = q;
      }
  }
  return PCG_OK;
}

int pcg_step(pcg_plan* p, const pcg_buffers* io, int32_t t, uint64_t seed, void* stream) {
  StepArgs a;
  int rc = fill_args(p, io, &a);
  if (rc != PCG_OK) return rc;
  if (io->B == 0) return PCG_OK;  // empty batch: nothing to do (zero-size buffers may be NULL)
  if (!io->x || !io->a || !io->obs || !io->rew || !io->done) return PCG_E_NULL;
  const DevConst& c = p->hc;
  if ((c.flags & PCG_F_A_DELTA) && !io->a_save) return PCG_E_NULL;
  a.t_scalar = t;
  a.seed = seed;
  const bool per_env_t = io->t != nullptr;
  const Kernels& k = kernels(p->model_id);
  const bool lds_st = p->lds_stages && p->integrator_id == PCG_INT_DOPRI5 && k.has_lds_stages;
  const int block = tb(lds_st);
  size_t shmem = lds_st ? sizeof(double) * 6 * (size_t)k.nx * BLOCK_LDS : 0;
  if (per_env_t) {
    const size_t sb = sizeof(double) * (size_t)(c.nsp + c.nd) * c.N;
    if (sb > 0 && shmem + sb <= 64 * 1024) {
      a.sched_in_lds = 1;
      shmem += sb;
    }
  }
  if (c.nunc > 0) {  // per-env uncertain parameters: dedicated general kernel
    if (!io->p_unc) return PCG_E_NULL;
    StepFn ufn = k.step_unc[p->integrator_id][per_env_t ? 1 : 0];
    if (!ufn) return PCG_E_UNSUPPORTED;
    size_t sh = 0;
    if (per_env_t) {
      const size_t sb = sizeof(double) * (size_t)(c.nsp + c.nd) * c.N;
      if (sb > 0 && sb <= 64 * 1024) {
        a.sched_in_lds = 1;
        sh = sb;
      }
    }
    hipLaunchKernelGGL(ufn, dim3(grid_for(io->B, BLOCK)), dim3(BLOCK), sh, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  // lean variant when no noise / Gaussian disturbance / constraint work is configured
  // (the lean kernels also compile out a_delta, the terminal "batch" reward and per-env disturbances)
  const bool extras = (c.flags & (PCG_F_NOISE | PCG_F_GAUSS_DIST | PCG_F_A_DELTA | PCG_F_REWARD_BATCH)) ||
                      c.ncon > 0 || io->d != nullptr;
  // streaming (persistent, prefetching, 16 B/lane) kernel for the lean lock-stepped path
  if (!per_env_t && !extras && !lds_st && !io->viol && p->variant != 1 && k.stream[p->integrator_id][0][0]) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool epl2_ok = k.stream[p->integrator_id][1][0] && (io->B % 2 == 0) && al16(io->x) && al16(io->a) &&
                         al16(io->obs) && al16(io->rew) && (reinterpret_cast<uintptr_t>(io->done) & 1u) == 0;
    int epl = (p->variant == 2) ? 1 : (epl2_ok ? 2 : 1);
    if (p->variant == 3 && !epl2_ok) return PCG_E_UNSUPPORTED;
    int lu = p->stream_unr;  // log2(sub-tiles per workgroup)
    if (lu < 0 || lu > 2 || !k.stream[p->integrator_id][epl - 1][lu]) lu = 0;
    StepFn sfn = k.stream[p->integrator_id][epl - 1][lu];
    // auto (0): the software-pipelined kernel where it exists (measured best on the cstr workload:
    // 14.9 us vs 15.0 two-sub-tile streaming vs 16.9 plain streaming vs 21 classic, profiles/r1)
    const bool piped = (p->variant == 4 || p->variant == 0) && p->integrator_id == PCG_INT_RK4 && k.pipe[epl - 1];
    if (piped) {
      sfn = k.pipe[epl - 1];
      lu = 0;
    }
    int& occ = piped ? p->pipe_occ[epl - 1] : p->stream_occ[epl - 1][lu];
    if (occ == 0) {
      const int q = resident_blocks(sfn);
      if (q < 0) return -q;
      occ = q;
    }
    const int64_t tile_envs = (int64_t)BLOCK * epl * (1 << lu);
    const int64_t ntile = (io->B + tile_envs - 1) / tile_envs;
    int bpc = occ;
    if (p->stream_bpc > 0 && p->stream_bpc < bpc) bpc = p->stream_bpc;
    int64_t grid = (int64_t)p->num_cus * bpc;
    if (grid > ntile) grid = ntile;
    a.nt_stores = p->nt_stores;
    hipLaunchKernelGGL(sfn, dim3((unsigned)grid), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  StepFn fn = k.step[p->integrator_id][per_env_t ? 1 : 0][lds_st ? 1 : 0][extras ? 1 : 0];
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(fn, dim3(grid_for(io->B, block)), dim3(block), shmem, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pcg_rollout_strided(pcg_plan* p, const pcg_buffers* io, int32_t t0, int32_t T, const double* a_seq,
                        int64_t a_step_stride, int64_t a_comp_stride, double* obs_seq, int64_t obs_step_stride,
                        int64_t obs_comp_stride, double* rew_seq, int64_t rew_step_stride, uint64_t seed,
                        void* stream) {
  StepArgs a;
  int rc = fill_args(p, io, &a);
  if (rc != PCG_OK) return rc;
  if (io->t) return PCG_E_UNSUPPORTED;  // lock-stepped only
  if (T < 1) return PCG_E_VALUE;
  if (io->B == 0) return PCG_OK;
  if (!io->x || !a_seq || !io->obs || !io->rew || !io->done) return PCG_E_NULL;
  const DevConst& c = p->hc;
  if ((c.flags & PCG_F_A_DELTA) && !io->a_save) return PCG_E_NULL;
  if (c.nunc > 0) return PCG_E_UNSUPPORTED;  // parameter uncertainty: per-step kernel only
  a.t_scalar = t0;
  a.seed = seed;
  a.T = T;
  a.a_seq = a_seq;
  a.obs_seq = obs_seq;
  a.rew_seq = rew_seq;
  a.a_ss = a_step_stride; a.a_cs = a_comp_stride;
  a.o_ss = obs_step_stride; a.o_cs = obs_comp_stride;
  a.r_ss = rew_step_stride;
  if (a.a_cs < io->B || (obs_seq && a.o_cs < io->B)) return PCG_E_DIM;
  const Kernels& k = kernels(p->model_id);
  const bool lds_st = p->lds_stages && p->integrator_id == PCG_INT_DOPRI5 && k.has_lds_stages;
  const bool extras = (c.flags & (PCG_F_NOISE | PCG_F_GAUSS_DIST | PCG_F_A_DELTA | PCG_F_REWARD_BATCH)) ||
                      c.ncon > 0 || io->d != nullptr;
  if (!extras && p->integrator_id == PCG_INT_RK4 && !io->viol && p->variant != 1 && k.roll_lean[0]) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool ev = ((a.a_ss | a.a_cs | a.o_ss | a.o_cs | a.r_ss) & 1) == 0;  // 16-byte rows stay 16-byte aligned
    const bool e2 = ev && k.roll_lean[1] && (io->B % 2 == 0) && al16(io->x) && al16(a_seq) && al16(io->obs) &&
                    al16(io->rew) && (!obs_seq || al16(obs_seq)) && (!rew_seq || al16(rew_seq)) &&
                    (reinterpret_cast<uintptr_t>(io->done) & 1u) == 0;
    const int epl = e2 ? 2 : 1;
    hipLaunchKernelGGL(k.roll_lean[epl - 1], dim3(grid_for(io->B, BLOCK * epl)), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  const int block = tb(lds_st);
  const size_t shmem = lds_st ? sizeof(double) * 6 * (size_t)k.nx * BLOCK_LDS : 0;
  StepFn fn = k.rollout[p->integrator_id][lds_st ? 1 : 0];
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(fn, dim3(grid_for(io->B, block)), dim3(block), shmem, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pcg_rollout(pcg_plan* p, const pcg_buffers* io, int32_t t0, int32_t T, const double* a_seq, double* obs_seq,
                double* rew_seq, uint64_t seed, void* stream) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!io) return PCG_E_NULL;
  const int64_t B = io->B;
  return pcg_rollout_strided(p, io, t0, T, a_seq, (int64_t)p->hc.na * B, B, obs_seq, (int64_t)p->hc.nobs * B, B, rew_seq,
                             B, seed, stream);
}

int pcg_reset(pcg_plan* p, const pcg_buffers* io, const uint8_t* mask, uint64_t seed, void* stream) {
  StepArgs a;
  int rc = fill_args(p, io, &a);
  if (rc != PCG_OK) return rc;
  if (io->B == 0) return PCG_OK;
  if (!io->x || !io->obs) return PCG_E_NULL;
  if (p->hc.nunc > 0 && !io->p_unc) return PCG_E_NULL;
  a.mask = mask;
  a.seed = seed;
  hipLaunchKernelGGL(reset_kernel, dim3(grid_for(io->B)), dim3(BLOCK), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// ---- step graph: T pcg_step launches recorded once, replayed with one host call ------------------------------
struct pcg_graph {
  uint32_t magic;
  int device;
  hipGraph_t graph;
  hipGraphExec_t exec;
  int n_nodes;
};
static constexpr uint32_t GRAPH_MAGIC = 0x50434747u;  // 'PCGG'

int pcg_graph_create(pcg_graph** out, pcg_plan* p, const pcg_buffers* io, const double* const* a_steps,
                     const double* const* d_steps, int32_t t0, int32_t T, uint64_t seed, int with_reset) {
  if (!out) return PCG_E_NULL;
  *out = nullptr;
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!io || !a_steps) return PCG_E_NULL;
  if (io->t) return PCG_E_UNSUPPORTED;
  if (T <= 0 || t0 < 0 || io->B <= 0) return PCG_E_DIM;
  for (int j = 0; j < T; ++j)
    if (!a_steps[j] || (d_steps && !d_steps[j])) return PCG_E_NULL;
  pcg_buffers b = *io;
  {
    const int wrc = warm_occupancy(p);  // launch geometry is queried lazily: do it outside the capture
    if (wrc != PCG_OK) return wrc;
  }
  hipStream_t cs;
  HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  int rc = PCG_OK;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    hipStreamDestroy(cs);
    return (int)e;
  }
  if (with_reset) rc = pcg_reset(p, &b, nullptr, seed, cs);
  for (int j = 0; j < T && rc == PCG_OK; ++j) {
    b.a = a_steps[j];
    b.d = d_steps ? d_steps[j] : nullptr;
    rc = pcg_step(p, &b, t0 + j, seed, cs);
  }
  e = hipStreamEndCapture(cs, &g);
  hipStreamDestroy(cs);
  if (rc != PCG_OK) {
    if (g) hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    hipGraphDestroy(g);
    return (int)e;
  }
  pcg_graph* q = new (std::nothrow) pcg_graph();
  if (!q) {
    hipGraphExecDestroy(ex);
    hipGraphDestroy(g);
    return PCG_E_VALUE;
  }
  q->magic = GRAPH_MAGIC;
  q->device = p->device;
  q->graph = g;
  q->exec = ex;
  q->n_nodes = T + (with_reset ? 1 : 0);
  *out = q;
  return PCG_OK;
}

int pcg_graph_launch(pcg_graph* q, void* stream) {
  if (!q || q->magic != GRAPH_MAGIC) return PCG_E_PLAN;
  return (int)hipGraphLaunch(q->exec, (hipStream_t)stream);
}

int pcg_graph_set_seed(pcg_graph* q, uint64_t seed) {
  if (!q || q->magic != GRAPH_MAGIC) return PCG_E_PLAN;
  size_t n = 0;
  HIP_TRY(hipGraphGetNodes(q->graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  HIP_TRY(hipGraphGetNodes(q->graph, nodes.data(), &n));
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType ty;
    HIP_TRY(hipGraphNodeGetType(nodes[i], &ty));
    if (ty != hipGraphNodeTypeKernel) continue;
    hipKernelNodeParams kp;
    HIP_TRY(hipGraphKernelNodeGetParams(nodes[i], &kp));
    if (!kp.kernelParams || !kp.kernelParams[0]) return PCG_E_UNSUPPORTED;
    // every kernel this library records takes one by-value StepArgs: re-key it in the graph and in the executable
    StepArgs a;
    std::memcpy(&a, kp.kernelParams[0], sizeof(a));
    a.seed = seed;
    void* argv[1] = {&a};
    kp.kernelParams = argv;
    kp.extra = nullptr;
    HIP_TRY(hipGraphKernelNodeSetParams(nodes[i], &kp));
    HIP_TRY(hipGraphExecKernelNodeSetParams(q->exec, nodes[i], &kp));
  }
  return PCG_OK;
}

int pcg_graph_destroy(pcg_graph* q) {
  if (!q) return PCG_OK;
  if (q->magic != GRAPH_MAGIC) return PCG_E_PLAN;
  hipGraphExecDestroy(q->exec);
  hipGraphDestroy(q->graph);
  q->magic = 0;
  delete q;
  return PCG_OK;
}

int pcg_rhs(pcg_plan* p, int64_t B, const double* x, const double* u, double* dx, void* stream) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!x || !u || !dx) return PCG_E_NULL;
  if (B <= 0) return B == 0 ? PCG_OK : PCG_E_DIM;
  const Kernels& k = kernels(p->model_id);
  hipLaunchKernelGGL(k.rhs, dim3(grid_for(B)), dim3(BLOCK), 0, (hipStream_t)stream, (CDevConst*)p->dC, B, p->cfg_nu, x, u, dx);
  return (int)hipGetLastError();
}

int pcg_integrate(pcg_plan* p, int64_t B, double* x, const double* u, int32_t* nsteps, void* stream) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!x || !u) return PCG_E_NULL;
  if (B <= 0) return B == 0 ? PCG_OK : PCG_E_DIM;
  const Kernels& k = kernels(p->model_id);
  const bool lds_st = p->lds_stages && p->integrator_id == PCG_INT_DOPRI5 && k.has_lds_stages;
  const int block = tb(lds_st);
  const size_t shmem = lds_st ? sizeof(double) * 6 * (size_t)k.nx * BLOCK_LDS : 0;
  IntKFn fn = k.integ[p->integrator_id][lds_st ? 1 : 0];
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(fn, dim3(grid_for(B, block)), dim3(block), shmem, (hipStream_t)stream, (CDevConst*)p->dC, B, p->cfg_nu, x,
                     u, nsteps);
  return (int)hipGetLastError();
}

// Host-only validation of a cfg (what pcg_plan_create would return before touching the
// device): lets the host logic be tested on machines without a GPU.
int pcg_cfg_validate(const pcg_env_cfg* cfg) {
  DevConst* d = new (std::nothrow) DevConst();
  if (!d) return (int)hipErrorOutOfMemory;
  int rc = build_devconst(cfg, d, nullptr);
  delete d;
  return rc;
}

}  // extern "C"
