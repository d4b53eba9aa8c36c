/*
 * pcgym_hip.h -- C ABI of libpcgym_hip.so, the MI355X (gfx950) batched
 * process-control environment engine.
 *
 * This is the drop-in boundary for pc-gym's per-timestep hot path.  In the
 * reference (pc-gym v0.1.8, all Python) that path is
 *
 *     make_env.step            src/pcgym/pcgym.py:350-500
 *       -> integration_engine.casadi_step / jax_step
 *                              src/pcgym/integrator.py:65-107, 163-182
 *       -> model.__call__      src/pcgym/model_classes.py:45-62, 370-412,
 *                                                        790-845, 891-913, 1272-1319
 *     make_env.reset           src/pcgym/pcgym.py:263-349
 *
 * one env per Python object, one CVODES object rebuilt per step.  Here the same
 * arithmetic runs for B environments per launch, one wavefront lane per
 * environment, over caller-owned SoA fp64 buffers  field[component][B].
 *
 * Conventions
 *   - plain C types only; no torch / C++ types cross this boundary.
 *   - every entry point returns an int status: 0 = ok, <0 = PCG_E_* (bad
 *     argument), >0 = a hipError_t.  Nothing throws, nothing calls exit().
 *   - all device buffers are caller-owned (e.g. torch tensors); the library
 *     allocates only the opaque plan (a few KB of device constants, the schedules and
 *     sample tables) and, on request, step-graph objects.
 *   - launches are asynchronous on the hipStream_t passed as `stream`
 *     (NULL = the default stream).  One plan may be driven by one host thread at
 *     a time; distinct plans / streams / devices are independent.
 *   - a plan belongs to the device that was current at pcg_plan_create().
 */
#ifndef PCGYM_HIP_H
#define PCGYM_HIP_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#else /* hipRTC (run-time compilation of the kernel headers with user expressions) ships no <stdint.h> */
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
typedef unsigned long uintptr_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define PCG_ABI_VERSION 14

#ifndef PCG_API
#define PCG_API __attribute__((visibility("default")))
#endif

/* capacity limits (compile-time; per-lane state lives in VGPRs) */
#define PCG_MAX_NX 24     /* physical states          */
#define PCG_MAX_NA 5      /* action dims              */
#define PCG_MAX_NDM 4     /* model disturbance inputs */
#define PCG_MAX_NSP 4     /* set-point keys           */
#define PCG_MAX_NCON 8    /* constraint rows          */
#define PCG_MAX_NUNC 8    /* uncertain parameters     */
#define PCG_MAX_PARAMS 128
#define PCG_MAX_NOBS (PCG_MAX_NX + PCG_MAX_NSP + PCG_MAX_NDM + PCG_MAX_NUNC)
#define PCG_MAX_NU (PCG_MAX_NA + PCG_MAX_NDM)
#define PCG_MAX_N 4096    /* episode length (schedule rows staged in LDS) */

/* status codes */
#define PCG_OK 0
#define PCG_E_NULL -1        /* required pointer is NULL                 */
#define PCG_E_MODEL -2       /* unknown model / integrator id            */
#define PCG_E_DIM -3         /* a dimension is out of range / mismatched */
#define PCG_E_VALUE -4       /* invalid scalar (dt<=0, substeps<1, ...)  */
#define PCG_E_PLAN -5        /* plan handle invalid / wrong device       */
#define PCG_E_UNSUPPORTED -6 /* combination not built                    */
#define PCG_E_JIT -7         /* a user expression did not compile (pcg_last_jit_log() has the compiler output) */

/* per-env health of a step, written to pcg_buffers.status (the reference's CVODES raises on an integration
 * failure, integrator.py:90-107; a batched kernel cannot raise per env, so it reports) */
#define PCG_ST_OK 0         /* the step was integrated over the full [0,dt] and the state is finite        */
#define PCG_ST_MAX_STEPS 1  /* DOPRI5: step budget (cfg.max_steps) exhausted before dt; state set to NaN    */
#define PCG_ST_UNDERFLOW 2  /* DOPRI5: step size underflow (blow-up / NaN right-hand side); state NaN      */
#define PCG_ST_NONFINITE 3  /* the integrator finished but the state is not finite (e.g. fixed-step RK4
                               outside its stability region, model_classes.py:1313 division by zero)     */

/* model ids: the reference registry keys (pcgym.py:128-148) that are on the hot path */
enum pcg_model {
  PCG_MODEL_CSTR = 0,          /* model_classes.py:23-62     nx=2  nu=1 (+Ti,Caf)      */
  PCG_MODEL_FOUR_TANK = 1,     /* model_classes.py:864-931   nx=4  nu=2                */
  PCG_MODEL_ME = 2,            /* model_classes.py:346-430   nx=10 nu=2 (+X0,Y6)       */
  PCG_MODEL_ME_REACTIVE = 3,   /* model_classes.py:763-861   nx=20 nu=2                */
  PCG_MODEL_CRYST = 4,         /* model_classes.py:1232-1345 nx=7  nu=1                */
  PCG_MODEL_AFFINE = 5,        /* custom_model whose RHS is affine: dx = A x + B u + c
                                  (pcgym.py:150-153; the reference's only KAT,
                                  tests/environment/test_make_env_custom_model.py:66-86); also carries the
                                  affine registry models hydraulic_tank :128-153, first_order_system :296-343,
                                  nonsmooth_control :509-558 (matrices built on the host) */
  /* "next" row f-2: further registry models (general kernels only, no streaming specialisation) */
  PCG_MODEL_COMPLEX_CSTR = 6,  /* model_classes.py:65-125   nx=4 nu=1 (+Ti,Caf) */
  PCG_MODEL_DISEASE = 7,       /* model_classes.py:156-183  nx=3 nu=1            */
  PCG_MODEL_BATCH = 8,         /* model_classes.py:222-265  nx=4 nu=1            */
  PCG_MODEL_PHOTO = 9,         /* model_classes.py:433-506  nx=3 nu=2 ("photobioreactor") */
  PCG_MODEL_CSTR_SERIES = 10,  /* model_classes.py:611-679  nx=4 nu=4            */
  PCG_MODEL_DISTILLATION = 11, /* model_classes.py:682-760  nx=9 nu=2            */
  PCG_MODEL_POLYMER = 12,      /* model_classes.py:1158-1229 nx=3 nu=4           */
  PCG_MODEL_BIOFILM = 13,      /* model_classes.py:1046-1155 nx=16 nu=5          */
  PCG_MODEL_HEAT_EX = 14,      /* model_classes.py:935-1044 nx=24 nu=4           */
  PCG_MODEL_INV_BATCH = 15,    /* model_classes.py:268-293 nx=4, no inputs (one ignored dummy action) */
  PCG_MODEL_OSCILLATORS = 16,  /* model_classes.py:186-216 nx=20 (N=10), no inputs (dummy action)     */
  PCG_MODEL_USER = 17,         /* custom_model with an arbitrary (non-affine) right-hand side (pcgym.py:150-153,
                                  tests/environment/test_make_env_custom_model.py:7-25): given as C source in
                                  pcg_env_cfg.user_rhs_src and compiled into this plan's kernels with hipRTC;
                                  nx <= PCG_MAX_NX, na <= PCG_MAX_NA, ndm <= PCG_MAX_NDM, n_params <= PCG_MAX_USER_PARAMS */
  PCG_MODEL_COUNT = 18
};
#define PCG_MAX_USER_PARAMS 64

/* integrators replacing integrator.py:90-107 (CVODES) / :65-88 (diffrax Tsit5) */
enum pcg_integrator {
  PCG_INT_RK4 = 0,     /* classical RK4, `substeps` equal sub-steps per env step (zero-order hold on u) */
  PCG_INT_DOPRI5 = 1,  /* adaptive Dormand-Prince 5(4), per-lane step size, rtol/atol
                          (mirrors integrator.py:61 PIDController(rtol=1e-8, atol=1e-8)) */
  PCG_INT_RODAS3 = 2,  /* stiff-capable: adaptive Rodas3 (4-stage linearly implicit Rosenbrock 3(2), L-stable; forward-
                          difference Jacobian, per-lane pivoted LU in LDS), rtol/atol/max_steps as for DOPRI5.  The
                          reference integrates with CVODES BDF (integrator.py:163-182).  Every model (nx <= 24);
                          pcg_step, pcg_step_autoreset, pcg_graph_* and pcg_integrate; pcg_rollout and plans with
                          per-env uncertain parameters return PCG_E_UNSUPPORTED */
  PCG_INT_RODAS4 = 3,  /* stiff-capable, fourth order: adaptive Rodas4 (Hairer & Wanner's RODAS: 6-stage linearly implicit
                          Rosenbrock 4(3) pair, gamma = 1/4, L-stable, stiffly accurate), same controller family as
                          Rodas3 (RMS norm, quantised factor 0.9 E^-1/4 in [0.2, 6]).  Linear algebra: models with a
                          structured analytic Jacobian (the 10-state extraction cascade: block-bidiagonal in both
                          directions) factor and solve W = I/(gamma h) - J in registers (6 reciprocals and ~230 flops per
                          step); every other model uses the forward-difference Jacobian and the per-lane pivoted LU in
                          LDS of Rodas3.  END-POINT ERROR CONTROL (cfg.ep_kmax > 0, models with a contraction-rate
                          hook): an env step only hands x(dt) on, and an error committed at time t is damped by
                          exp(-mu (dt - t)) on its way there, so the local tolerance of an attempted step ending at t'
                          is multiplied by 2^k, k = min(ep_kmax, floor(ep_frac mu log2(e) (dt - t'))) -- exact powers of
                          two, bit-identical in kernel and oracle.  The reference integrates with CVODES BDF
                          (integrator.py:163-182).  pcg_step, pcg_step_autoreset, pcg_graph_*, pcg_integrate;
                          pcg_rollout and per-env uncertain parameters: PCG_E_UNSUPPORTED */
  PCG_INT_TSIT5 = 4,   /* Tsitouras 5(4), FSAL: the method of the reference's jax path (integrator.py:56-61, diffrax.Tsit5 with
                          PIDController(rtol = atol = 1e-8)); controller, norm, initial step and failure semantics of
                          PCG_INT_DOPRI5.  General kernel (both counter modes), pcg_step_autoreset, pcg_graph_*,
                          pcg_integrate; pcg_rollout and per-env uncertain parameters: PCG_E_UNSUPPORTED */
  PCG_INT_RK4G = 5,    /* GUARDED RK4: `substeps` equal RK4 sub-steps, accepted per env only while the model's guard says the
                          fixed step is accurate -- no growing mode (largest growth rate g <= 0) and a resolved fastest
                          rate (rho h <= 1) at every sub-step start and at the end state, finite result; an env that
                          fails the guard is re-integrated from its start state by PCG_INT_DOPRI5 at rtol / atol before
                          pcg_step returns its launches to the stream -- inside the same kernel, or, for batches that fill
                          the chip, by a second launch of the adaptive pair's work-queue kernel over exactly the envs the
                          first one marked (same arithmetic per env, same bits; `done` holds the mark 2 only between the
                          two: a consumer of `done` on ANOTHER stream must order itself behind the whole pcg_step, as for any
                          output; should the second launch fail to start, pcg_step returns its error and the marked envs keep
                          done == 2 with their state, observation and counters of the step not advanced) -- (nsteps reports (0,0) for accepted envs, the pair's counts otherwise).  Models
                          with a guard hook only (cstr: the ignition branch).  The guard is the ONLY test -- RK4 carries no
                          error estimate -- and it is calibrated on the cstr's observation box and action box: an opt-in
                          for that box (the model's default plan is PCG_INT_T5G, which also checks an estimate).  General
                          kernel, pcg_step_autoreset, pcg_graph_*, pcg_integrate, pcg_rollout; not with per-env uncertain
                          parameters */
  PCG_INT_T5G = 6,     /* GUARDED FIXED-STEP TSIT5: `substeps` equal steps of Tsit5's fifth-order solution weights, TRUSTED per
                          env while (i) the model's guard holds at the START and END state of every step (g <= 0, rho h <= 2; the five
                          inner stage states change no decision once the estimate of (ii) is checked)
                          and (ii) the pair's own embedded 5(4) error estimate of every step stays below 4e-7 |x| + 4e-9 (RMS;
                          the seventh stage it needs is the next step's first stage and the end-state guard: no extra
                          evaluation); otherwise PCG_INT_DOPRI5 from the start state at rtol / atol, in one launch or two as
                          PCG_INT_RK4G (same nsteps convention).  Calibrated so that the canonical cstr loop is never
                          escalated and every trusted env of a wide state / input / step-size box lies inside 3 x the
                          reference's CVODES tolerances (1e-6 |x| + 1e-8) of the true solution.  12 + 1 evaluations per
                          canonical cstr step: the default plan of the cstr.  Models with a guard hook only; general kernel,
                          pcg_step_autoreset, pcg_graph_*, pcg_integrate, pcg_rollout; not with per-env uncertain
                          parameters */
  PCG_INT_CV8 = 7,     /* Cooper & Verner's explicit Runge-Kutta method of order 8 (11 stages), `substeps` equal steps per
                          env step: for smooth right-hand sides one step replaces several RK4 steps (four_tank's default
                          plan: 11 evaluations instead of 20 at a smaller error).  Kernels of PCG_INT_RK4: lean pipelined
                          kernel (small models), general kernel, pcg_step_autoreset, pcg_graph_*, pcg_integrate,
                          pcg_rollout; not with per-env uncertain parameters */
  PCG_INT_RODAS5 = 8,  /* stiff-capable, fifth order (ABI 13): adaptive Rodas5 (Di Marzo's coefficient set, the one of Hairer &
                          Wanner's RODAS5 code: 8-stage linearly implicit Rosenbrock 5(4) pair, gamma = 0.19, L-stable, stiffly
                          accurate).  Everything but the tableau and the controller's exponent (quantised factor 0.9 E^-1/5 in
                          [0.2, 6]) is PCG_INT_RODAS4's: linear-algebra policy (structured W in registers / dense W in LDS),
                          error norm, END-POINT ERROR CONTROL (cfg.ep_frac / ep_kmax), cooperative rule (cfg.coop_thr), first
                          step, failure semantics, entry points.  The extraction cascade is accuracy-bound under the fourth-
                          order pair (17.6 attempts per env step for 1e-6 of a 1e-13 solve); this pair reaches the same class in
                          0.6 x the attempts at 8 stages against 6: the default plan of multistage_extraction.  The reference
                          integrates with CVODES BDF, variable order up to 5 (integrator.py:163-182) */
  PCG_INT_COUNT = 9
};

/* cfg.flags */
#define PCG_F_NORMALISE_A 0x0001u   /* pcgym.py:59,372-375                                     */
#define PCG_F_NORMALISE_O 0x0002u   /* pcgym.py:60,483-489                                     */
#define PCG_F_A_DELTA 0x0004u       /* pcgym.py:57,376-383                                     */
#define PCG_F_R_PENALTY 0x0008u     /* pcgym.py:121,556-557 / 529-530                          */
#define PCG_F_DONE_ON_CONS 0x0010u  /* pcgym.py:120,613-614                                    */
#define PCG_F_NOISE 0x0020u         /* pcgym.py:453-466 multiplicative Gaussian obs noise      */
#define PCG_F_REWARD_BATCH 0x0040u  /* terminal reward, pcgym.py:502-532 (else SP reward)      */
#define PCG_F_MAXIMISE 0x0080u      /* batch reward sign, pcgym.py:524-527                     */
#define PCG_F_REF_COMPAT 0x0100u    /* replicate reference quirks Q1 (double action
                                       de-normalisation when normalise_a && a_delta,
                                       pcgym.py:372-379) and Q3 (constraint rows see a
                                       re-"de-normalised" state/input, pcgym.py:597-608)       */
#define PCG_F_GAUSS_DIST 0x0200u    /* extension: d = d_sched + d_sigma*z, z~N(0,1) Philox,
                                       clipped to [d_clip_lo,d_clip_hi] (BASELINE configs[4])  */
#define PCG_F_X0_NORMAL 0x0400u     /* reset-time x0 / parameter uncertainty is normal (else uniform),
                                       pcgym.py:255-261                                        */
#define PCG_F_UNC_EMPIRICAL 0x0800u /* uncertain parameters are drawn uniformly from per-parameter sample
                                       tables (env_params["empirical_distribution"], pcgym.py:311-316:
                                       np.random.choice) instead of value*(1 +- pct)           */
#define PCG_MAX_EMP 65536           /* total empirical samples over all parameters            */
#define PCG_F_REWARD_TRACK 0x1000u  /* declarative form of the custom_reward family every paper script uses
                                       (pc-gym_paper/train_policies/cstr/custom_reward.py:3-39, 4tank_train.py:16-52,
                                       me_train.py:17-53, Biofilm/biofilm_train.py:13-41, constraint_showcase/
                                       custom_reward.py:6-69):  r = -( sum_k r_scale_k ((o_k - SP_k[t]) / (hi_k - lo_k))^2
                                       + sum_j [ R_du (du_j / (a_hi_j - a_lo_j))^2 + R_u ((u_j - a_lo_j) / (a_hi_j - a_lo_j))^2 ]
                                       + [violated] sum_c box_c^2 )  on the (noisy) physical observation o and the
                                       physical action u, with the previous action kept per env in u_prev         */
#define PCG_F_REWARD_CRYST 0x2000u  /* with PCG_F_REWARD_TRACK on the crystallisation model: the tracked CV and Ln are
                                       recomputed from the observed moments, CV = sqrt(mu2 mu0 / mu1^2 - 1), Ln = mu1/mu0
                                       (pc-gym_paper/train_policies/crystalisation/cryst_train.py:17-48)                 */
#define PCG_MAX_RBOX 4              /* state boxes of the constraint-violation term                               */

/*
 * Environment configuration: the numeric content of the reference's
 * env_params dict (pcgym.py:32-253).  All arrays are small HOST arrays, copied
 * at pcg_plan_create(); pointers may be NULL when the matching count is 0.
 *
 * Layout of one env's "state"/observation vector, as in the reference
 * (pcgym.py:160-165,291-298,409-410,432-438):
 *      [ x(0..nx) | SP slot (nsp_obs) | configured disturbances (nd) | uncertain parameters (nunc) ]
 *                                                                   Nobs = nx+nsp_obs+nd+nunc
 *      (reset order of the reference, pcgym.py:291-316; with disturbances AND parameter uncertainty the
 *      reference's step() writes the disturbance slots at a different offset -- quirk Q11: this layout is
 *      kept in step() too, see d_param_index)
 * and of the model input vector (pcgym.py:371,386-404):
 *      uk = [ action (na) | model disturbance inputs (ndm) ]            Nu = na+ndm
 */
typedef struct pcg_env_cfg {
  int32_t model_id;       /* enum pcg_model */
  int32_t integrator_id;  /* enum pcg_integrator */
  int32_t nx;             /* physical states (reference: Nx_oracle)                     */
  int32_t na;             /* action dims (len(a_space.low))                             */
  int32_t ndm;            /* len(model.info()["disturbances"]) if disturbances active, else 0 */
  int32_t nd;             /* configured disturbance keys (reference: Nd), <= ndm         */
  int32_t nsp;            /* len(SP)                                                    */
  int32_t nsp_obs;        /* SP slots present in the state/obs vector: nsp when x0 carries them
                             (len(x0) == nx+nsp, the documented form), 0 when x0 has only the nx
                             physical states -- the reference then silently drops the SP slot
                             (pcgym.py:438 assigns into an empty slice), as in its own KAT      */
  int32_t ncon;           /* constraint rows (reference: n_con)                         */
  int32_t nrew;           /* batch reward: number of reward states                      */
  int32_t nunc;           /* uncertain model parameters sampled per env at reset (pcgym.py:301-310) */
  int32_t N;              /* episode length (reference: N); done when t == N-1          */
  int32_t substeps;       /* RK4 sub-steps per env step (>=1)                           */
  int32_t max_steps;      /* DOPRI5: step budget per env step (accepted+rejected)       */
  uint32_t flags;         /* PCG_F_*                                                    */
  int32_t n_params;
  double dt;              /* tsim / N  (pcgym.py:110)                                   */
  double rtol, atol;      /* DOPRI5 tolerances                                          */

  const double* params;   /* [n_params] model parameters, order = pcg_model_param_names() */
  const double* x0;       /* [nx+nsp_obs] reference x0 incl. SP slots (pcgym.py:108,284) */
  const double* x0_unc;   /* [nx] or NULL: reset-time x0 uncertainty fraction (pcgym.py:285-288) */
  const double* a_low;    /* [na] a_space                                               */
  const double* a_high;   /* [na]                                                       */
  const double* a_act_low;   /* [na] a_space_act (a_delta clip), pcgym.py:383           */
  const double* a_act_high;  /* [na]                                                    */
  const double* a_0;      /* [na] a_delta initial action (pcgym.py:58,320)              */
  const double* o_low;    /* [Nobs] observation_space_base incl. disturbance bounds     */
  const double* o_high;   /* [Nobs]                                                     */
  const uint8_t* obs_mask;/* [nx] or NULL: 1 = observed (partial_observation)           */
  const int32_t* sp_index;/* [nsp] state index of each SP key (pcgym.py:553)            */
  const double* sp;       /* [nsp][N] set-point schedules                               */
  const double* r_scale;  /* [nsp] (SP reward) or [nrew] (batch reward)                 */
  const int32_t* rew_index;  /* [nrew] batch reward state indices                       */
  const int32_t* d_slot;  /* [nd] for each configured key, its index in the model's
                             disturbance list (ascending; pcgym.py:392-398)             */
  const double* d_sched;  /* [nd][N] disturbance schedules (pcgym.py:173,394)           */
  const double* d_default;/* [ndm] model default for unconfigured inputs (pcgym.py:400-404) */
  const double* d_sigma;  /* [nd] PCG_F_GAUSS_DIST                                      */
  const double* d_clip_lo;/* [nd]                                                       */
  const double* d_clip_hi;/* [nd]                                                       */
  const double* con_A;    /* [ncon][Nobs+Nu] affine constraint rows g = A.[state;uk] - b <= 0,
                             the declarative form of the reference's callable g(x,u)
                             (pcgym.py:560-577, docs/guides/constraints.md:35-51)       */
  const double* con_b;    /* [ncon]                                                     */
  const double* noise_pct;/* [nx] per-state noise fraction (pcgym.py:454-466)           */
  const int32_t* unc_index; /* [nunc] index of each uncertain parameter in `params`       */
  const double* unc_pct;  /* [nunc] uncertainty fraction (uniform half-width / normal sigma, pcgym.py:255-261) */
  const double* unc_emp;  /* PCG_F_UNC_EMPIRICAL: concatenated sample tables, parameter j owns
                             unc_emp[unc_emp_off[j] .. unc_emp_off[j+1])                   */
  const int32_t* unc_emp_off; /* [nunc+1], unc_emp_off[0] = 0                              */
  /* PCG_F_REWARD_TRACK */
  double rew_R_du;        /* weight of the squared normalised action increment (R in custom_reward.py:6)     */
  double rew_R_u;         /* weight of the squared normalised action itself (biofilm_train.py:16,39)         */
  int32_t rew_nbox;       /* state boxes penalised while a constraint row is violated (0..PCG_MAX_RBOX)      */
  const int32_t* rew_box_index; /* [rew_nbox] state index                                                     */
  const double* rew_box_lo;     /* [rew_nbox] physical lower bound (constraint_showcase/custom_reward.py:4-5) */
  const double* rew_box_hi;     /* [rew_nbox] physical upper bound                                            */
  /* User expressions: the NON-affine form of the reference's callables (constraints(x,u) pcgym.py:119-125, 560-577;
   * custom_reward(self, obs, uk, violated) pcgym.py:201-205, 470-471), given as C source and compiled into this plan's
   * step kernel with hipRTC at pcg_plan_create() (cached by source hash: in the process and under $PCG_JIT_CACHE).
   *   user_cons_src    statements that fill g[0 .. ncon-1] (double) from  x[] = the reference's state vector
   *                    [x | SP slots | disturbances] (physical units) and u[] = uk [action | disturbance inputs];
   *                    con_A / con_b are ignored when it is given.  Quirk Q3 (state / input "de-normalised" once
   *                    more under PCG_F_REF_COMPAT with normalisation on) is applied to x[] and u[] first, as for rows.
   *   user_reward_src  ONE expression of type double over  o[] = the (noisy) physical observation vector the reference
   *                    hands to custom_reward, x[] = the noise-free state vector, u[] = uk, sp[] = SP_k[t] at the new t,
   *                    violated (0/1), t (new step counter), N.  Replaces the built-in reward.
   * Available in the one-env-per-lane general kernel only (any integrator, lock-stepped or per-env counters) and in
   * pcg_rollout (the run-time compiled module carries its own fused rollout kernel; not for the Rosenbrock integrators,
   * whose matrices live in LDS); not with per-env uncertain parameters or pcg_graph.  Math: exp log sqrt pow fabs fmin
   * fmax sin cos tanh. */
  const char* user_cons_src;
  const char* user_reward_src;
  const char* jit_include_dir;  /* directory holding pcg_kernels.hpp and its siblings (the library's own csrc/)      */
  /* PCG_MODEL_USER: the model's right-hand side (the reference's custom_model.__call__(x, u), pcgym.py:150-153) as C
   * statements that fill dx[0 .. nx-1] (double) from x[] (nx states), u[] (na inputs, then ndm disturbance inputs) and
   * p[] (n_params parameters = cfg.params).  Compiled with hipRTC into this plan's general step kernel (both time
   * modes), pcg_integrate, pcg_rhs and pcg_rollout (the last not for PCG_INT_RODAS3 / PCG_INT_RODAS4 / PCG_INT_RODAS5); any integrator;
   * composes with user_cons_src / user_reward_src.  Not available: per-env uncertain parameters. */
  const char* user_rhs_src;
  /* PCG_INT_RODAS4 / PCG_INT_RODAS5, end-point error control (see enum pcg_integrator): 0 / 0 = classical local error control */
  double ep_frac;         /* fraction of the model's contraction rate credited to the damping (0.5 by default: the cascade
                             is non-normal -- a perturbation travels down the stages before it decays)                 */
  int32_t ep_kmax;        /* largest exponent: tolerances are relaxed by at most 2^ep_kmax (0 = off; the Python side's defaults:
                             10 under PCG_INT_RODAS4, 16 under PCG_INT_RODAS5, whose attempts additionally cap the exponent
                             at 2 bits per remaining step of their size, trunc(2 (dt - t') / h): a Rosenbrock step damps a
                             stiff component by |R(h lambda)| ~ 0.1-0.16 only, whatever exp(h lambda) says)              */
  /* Disturbances TOGETHER with per-env uncertain parameters (pcgym.py:291-316, 386-412; quirk Q11).  The state /
   * observation layout is the reference's reset() order [x | SP | d | unc] in reset AND step (its step() writes the
   * disturbance slots at another offset -- the one place where this engine deliberately does not follow it); a model
   * disturbance input that is NOT configured takes the env's own (possibly uncertain) parameter value, as the
   * reference's `self.model.info()["parameters"][k]` does (pcgym.py:400-404).  d_param_index names that parameter. */
  const int32_t* d_param_index; /* [ndm] or NULL: index in `params` of each model disturbance input                  */
  /* PCG_INT_RODAS4 / PCG_INT_RODAS5 on a model with a cooperative rule (multistage_extraction with eq_exponent == 2): env steps whose
   * predicted cost -- the model's fit of the pair's attempts per env step from the held input and the scaled size of
   * f(x0), in exact arithmetic -- reaches coop_thr are integrated by SEULEX-8 (extrapolated linearly implicit Euler, fixed
   * column of eight, same accuracy class: pcg_seulex.hpp) instead of the pair.  In the work-queue kernel eight lanes share
   * such an env (one row of the extrapolation tableau each), elsewhere one lane runs the eight rows: the same bits either
   * way (the threshold counts attempts of the FOURTH-order pair under either).  A launch is as long as its heaviest env; this is what shortens it (the reference's CVODES integrates a stiff
   * column at a cost that does not depend on its batch-mates, integrator.py:163-182).  0 = off (every env takes the pair);
   * nsteps then counts big steps for the heavy envs.  PCG_E_UNSUPPORTED for other models / integrators when > 0. */
  double coop_thr;
} pcg_env_cfg;

/*
 * Per-call device buffers (caller-owned, SoA, fp64 unless noted).
 * "in/out" buffers are updated in place.
 */
typedef struct pcg_buffers {
  int64_t B;          /* environments in this launch                                            */
  double* x;          /* [nx][B]    in/out  physical state                                      */
  const double* a;    /* [na][B]    in      policy action (normalised if PCG_F_NORMALISE_A)     */
  const double* d;    /* [nd][B]    in|NULL per-env explicit disturbance values for this step;
                                            NULL = use the plan's shared schedule               */
  int32_t* t;         /* [B]        in/out|NULL per-env step counter; NULL = lock-stepped batch,
                                            the scalar `t` argument is used                     */
  double* a_save;     /* [na][B]    in/out|NULL a_delta accumulator (required with PCG_F_A_DELTA) */
  double* obs;        /* [Nobs][B]  out     observation                                         */
  double* rew;        /* [B]        out     reward                                              */
  uint8_t* done;      /* [B]        out     episode finished                                    */
  uint8_t* viol;      /* [B]        out|NULL any constraint row > 0 after the step              */
  double* g;          /* [ncon][B]  out|NULL constraint rows after the step (cons_info[:,t,:])  */
  double* g_pre;      /* [ncon][B]  out|NULL rows of the pre-step check the reference runs when
                                            t==0 (pcgym.py:416-420); untouched for other t      */
  int32_t* nsteps;    /* [2][B]     out|NULL DOPRI5 accepted / rejected step counts             */
  double* u_prev;     /* [na][B]    in/out|NULL previous physical action (required with PCG_F_REWARD_TRACK);
                                            NaN = "no previous action yet" (first step: du = 0, the reference's
                                            hasattr(self, 'u_prev') branch, custom_reward.py:7-8); pcg_reset
                                            leaves it alone, as the reference never clears u_prev              */
  double* p_unc;      /* [nunc][B]  in/out  per-env values of the uncertain parameters: written by
                                            pcg_reset, read by pcg_step (required when nunc > 0)        */
  uint8_t* status;    /* [B]     in/out|NULL per-env health, STICKY: a step that is not PCG_ST_OK writes its
                                            PCG_ST_* code, an OK step leaves the entry alone -- the buffer holds "what
                                            went wrong since the host last cleared it" and costs no traffic while
                                            nothing does.  An env whose adaptive integration fails gets a NaN state
                                            (never a silently wrong one) whether or not this buffer is given      */
} pcg_buffers;

typedef struct pcg_plan pcg_plan; /* opaque */

/* library / ABI version (PCG_ABI_VERSION). */
PCG_API int pcg_version(void);
/* Build id: a digest of the library's sources (csrc: the kernel headers and the .hip units) and of this header at build time (the Makefile's PCG_SRC_HASH;
   "unknown-build" for a build made without it).  Measurements that cannot be taken in-process -- the hardware-counter
   passes under profiles/ -- record it, and bench.py reports their traffic figures only for the build they were taken on. */
PCG_API const char* pcg_build_id(void);

/* human-readable text for a status returned by any entry point (static storage). */
PCG_API const char* pcg_strerror(int status);

/* model metadata = model.info() of the reference (model_classes.py:8-20, 414-430, ...):
 * fills nx, nu (inputs), ndm (disturbance inputs), n_params.  */
PCG_API int pcg_model_info(int model_id, int32_t* nx, int32_t* nu, int32_t* ndm, int32_t* n_params);

/* default parameter vector of a model, in the order the kernels expect
 * (= declaration order of the reference dataclass fields).  out has n_params slots. */
PCG_API int pcg_model_default_params(int model_id, double* out, int32_t n_out);

/* Build a plan: validates cfg, folds the affine maps (action de-normalisation,
 * observation normalisation, compat transforms) and uploads constants and
 * schedules to the current device.  Replaces integration_engine.__init__
 * (integrator.py:19-63) + make_env._setup_* numeric state (pcgym.py:56-253). */
PCG_API int pcg_plan_create(pcg_plan** out, const pcg_env_cfg* cfg);
PCG_API int pcg_plan_destroy(pcg_plan* plan);

/* Algorithmic HBM bytes one env-step moves for this plan and buffer set
 * (SURVEY.md section 8d formula); used by bench.py for the roofline line. */
PCG_API int64_t pcg_plan_bytes_per_env_step(const pcg_plan* plan, const pcg_buffers* io);

/* One fused environment step for B envs: replaces make_env.step (pcgym.py:350-500):
 * action map -> disturbance injection -> ODE integration over [0,dt] -> SP slot ->
 * constraint rows -> done -> observation noise -> reward -> observation normalisation.
 * `t` is the pre-step counter for lock-stepped batches (io->t == NULL).
 * `seed` keys the counter-based RNG (Philox4x32-10 on (seed, env index, t)). */
PCG_API int pcg_step(pcg_plan* plan, const pcg_buffers* io, int32_t t, uint64_t seed, void* stream);

/* Reset (pcgym.py:263-349): x <- x0 (with optional x0 uncertainty), t <- 0,
 * a_save <- a_0, obs <- normalised [x0 | SP slots of x0 | d[:,0]].
 * mask [B] u8 or NULL: only envs with mask!=0 are reset (masked auto-reset).
 * env_offset: global index of env 0 of this shard (keys the RNG; multi-GPU). */
PCG_API int pcg_reset(pcg_plan* plan, const pcg_buffers* io, const uint8_t* mask, uint64_t seed,
              void* stream);

/* Global env index of local env 0 (batch sharded over GPUs); default 0. */
PCG_API int pcg_plan_set_env_offset(pcg_plan* plan, int64_t env_offset);

/* Tuning / sharding options. */
#define PCG_OPT_ENV_OFFSET 1  /* same as pcg_plan_set_env_offset                              */
#define PCG_OPT_LDS_STAGES 2  /* DOPRI5: keep stage vectors k1..k6 in LDS [stage][comp][lane]
                                 (64-thread workgroups) instead of VGPRs; default 0           */
#define PCG_OPT_VARIANT 3     /* step-kernel selection: 0 auto (default), 1 classic one-env-per-lane
                                 grid, 2 streaming persistent kernel 1 env/lane, 3 streaming 2 envs/lane
                                 (16 B per lane accesses), 4 software-pipelined lean kernel; 5 = the in-workgroup
                                 WORK QUEUE for an adaptive plan of any model (by default only models with a cost
                                 key, the extraction columns, go through it): pays when the step counts of a batch
                                 are heavy-tailed -- cstr envs on the ignition branch under PCG_INT_DOPRI5: 572 ->
                                 451 us per 2^20-env step; costs when they are not (canonical cstr loop: 38 -> 90 us).
                                 1-4 are for A/B measurement                                                     */
#define PCG_OPT_STREAM_BLOCKS_PER_CU 4 /* streaming kernel: resident workgroups per CU (0 = occupancy query) */
#define PCG_OPT_NT_STORES 5   /* streaming kernel: non-temporal stores for obs / reward (not re-read by the step) */
PCG_API int pcg_plan_set_option(pcg_plan* plan, int option, int64_t value);

/* Host-only validation of a cfg: the status pcg_plan_create() would return before it
 * touches the device (usable on machines without a GPU). */
PCG_API int pcg_cfg_validate(const pcg_env_cfg* cfg);

/* Test hook: dx = f(x,u) of the plan's model for B envs.
 * x [nx][B], u [Nu][B] (physical units), dx [nx][B]. */
PCG_API int pcg_rhs(pcg_plan* plan, int64_t B, const double* x, const double* u, double* dx, void* stream);

/* Test hook: integrate only.  x [nx][B] in/out, u [Nu][B] physical, held for [0,dt]
 * (zero-order hold, integrator.py:163-182).  nsteps [2][B] or NULL. */
PCG_API int pcg_integrate(pcg_plan* plan, int64_t B, double* x, const double* u, int32_t* nsteps,
                  void* stream);

/* Open-loop fused rollout ("next" row f-1: counterpart of policy_eval.rollout,
 * policy_evaluation.py:71-130): T env steps with the state kept in registers.
 * a_seq [T][na][B]; obs_seq [T][Nobs][B]|NULL; rew_seq [T][B]|NULL; io->obs/rew/done
 * receive the last step.  Lock-stepped only (io->t must be NULL). */
PCG_API int pcg_rollout(pcg_plan* plan, const pcg_buffers* io, int32_t t0, int32_t T, const double* a_seq,
                double* obs_seq, double* rew_seq, uint64_t seed, void* stream);

/* pcg_rollout with explicit element strides (step, component) for the three sequences; the env index is
 * unit-stride.  Lets the collector write straight into the reference's axis order
 * x (Nx, N, reps), u (Nu, N, reps), r (1, N, reps)  (policy_evaluation.py:155-197):
 * obs_comp_stride = N*B, obs_step_stride = B.  Strides must keep rows 16-byte aligned for the 2-env/lane path. */
PCG_API int pcg_rollout_strided(pcg_plan* plan, const pcg_buffers* io, int32_t t0, int32_t T, const double* a_seq,
                                int64_t a_step_stride, int64_t a_comp_stride, double* obs_seq,
                                int64_t obs_step_stride, int64_t obs_comp_stride, double* rew_seq,
                                int64_t rew_step_stride, uint64_t seed, void* stream);

/* pcg_step followed, in the same launch, by the reset of every env that finished in it (gymnasium "same-step"
 * auto-reset: rew / done / viol are those of the finished step; x, obs, t, a_save and the per-env parameters are
 * those of the new episode, drawn with `reset_seed`).  Equivalent to pcg_step + pcg_reset(mask = io->done,
 * seed = reset_seed) -- the reference has no counterpart (its callers loop "if done: env.reset()",
 * policy_evaluation.py:86-128) -- but one launch instead of two.  With per-env step counters (io->t) `t` is
 * ignored; for a lock-stepped batch `t` is the shared counter and the caller restarts it at 0 after the call
 * that returns done (t == N-2, or a constraint violation with PCG_F_DONE_ON_CONS ends single envs early). */
PCG_API int pcg_step_autoreset(pcg_plan* plan, const pcg_buffers* io, int32_t t, uint64_t seed, uint64_t reset_seed,
                               void* stream);

/* Step graph: T consecutive pcg_step launches (t = t0 .. t0+T-1, optionally preceded by a full pcg_reset)
 * recorded once as a HIP graph and replayed with ONE host call.  This is the on-device form of the
 * reference's per-episode Python loop "for i in range(N-1): env.step(a_i)" (policy_evaluation.py:86-128)
 * for callers that still want every step's obs/rew in the plan's buffers between launches: same kernels,
 * same buffers, no launch-to-launch gap (measured 15.0 -> 13.6 us per step on the 2^20-env cstr workload).
 * a_steps [T] device pointers, each [na][B] (entries may repeat); d_steps NULL or [T] pointers, each [nd][B].
 * t0, T, seed and all buffer addresses are baked into the graph; pcg_graph_set_seed() re-keys the RNG of an
 * instantiated graph (new episode, fresh noise) without re-recording.  io->t must be NULL (lock-stepped). */
typedef struct pcg_graph pcg_graph;
PCG_API int pcg_graph_create(pcg_graph** out, pcg_plan* plan, const pcg_buffers* io, const double* const* a_steps,
                             const double* const* d_steps, int32_t t0, int32_t T, uint64_t seed, int with_reset);
PCG_API int pcg_graph_launch(pcg_graph* graph, void* stream);
PCG_API int pcg_graph_set_seed(pcg_graph* graph, uint64_t seed);
PCG_API int pcg_graph_destroy(pcg_graph* graph);

/* Compiler output of the most recent failed run-time compilation in this process (static storage, "" if none). */
PCG_API const char* pcg_last_jit_log(void);

/* Test hook: the work-queue kernel's tile sort on its own.  words: ntiles x S 32-bit words on the device (distinct
 * within a tile), sorted in place, DESCENDING, one workgroup of `threads` threads per tile.  (S, threads) as the step kernels
 * use them: S in {512, 1024, 2048}, threads in {256, 512}; anything else PCG_E_UNSUPPORTED.  The step results do not depend
 * on the order of a tile -- this is the only way to see that the sort sorts. */
PCG_API int pcg_test_sort_tile(uint32_t* words, int32_t S, int32_t threads, int64_t ntiles, void* stream);

/* Kernel-instantiation coverage (TEST HOOK; active only when PCG_COVERAGE is set in the environment at load time, otherwise
 * returns -1 and records nothing).  Every kernel launch of this library notes the instantiation it launches; this call
 * writes their MANGLED names, newline-separated and NUL-terminated, into buf (at most cap bytes; run-time compiled kernels
 * carry the prefix "jit:"), and returns the size the full list needs.  reset != 0 empties the record afterwards.
 * tests/conftest.py asks after every GPU test; tools/kernel_inventory.py lists what the library carries. (host) */
PCG_API int64_t pcg_coverage_names(char* buf, int64_t cap, int reset);

/* Raw Philox4x32-10 block for KAT tests: ctr[4], key[2] -> out[4]. (host) */
PCG_API void pcg_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* PCGYM_HIP_H */
