// pcg_abi.hip -- the C ABI of include/pcgym_hip.h: configuration folding (build_devconst), plan / graph
// objects, host-side kernel dispatch, and the model-independent reset kernel.  The kernel templates
// live in pcg_kernels.hpp; each model's instantiations are compiled in a pcg_inst_*.hip unit.
#include "pcg_kernels.hpp"
#include "pcg_step_feat.hpp"

#include <hip/hiprtc.h>
#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>

#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <string>

namespace pcg {

// reset (pcgym.py:263-349)
__global__ __launch_bounds__(BLOCK) void reset_kernel(const StepArgs A) {
  CDevConst& c = *A.C;
  const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= A.B) return;
  if (A.mask && !A.mask[e]) return;
  reset_env(A, c, e, A.seed);
}

// one table of kernel instantiations per model, built in the pcg_inst_*.hip units
Kernels kernels_cstr(), kernels_four_tank(), kernels_me(), kernels_me_reactive(), kernels_cryst(), kernels_affine();
Kernels kernels_complex_cstr(), kernels_disease(), kernels_batch(), kernels_photo(), kernels_cstr_series();
Kernels kernels_distillation(), kernels_polymer(), kernels_biofilm(), kernels_heat_ex(), kernels_inv_batch();
Kernels kernels_oscillators(), kernels_me_sq(), kernels_me_reactive_sq();

// PCG_MODEL_USER has no ahead-of-time kernels: every launch of such a plan goes through its run-time compiled module
static Kernels kernels_user_stub() {
  Kernels k;
  std::memset(&k, 0, sizeof(k));
  k.nraw = -1;
  return k;
}

static const Kernels& kernels(int id) {
  static const Kernels K[PCG_KID_COUNT] = {
      kernels_cstr(),        kernels_four_tank(),   kernels_me(),      kernels_me_reactive(), kernels_cryst(),
      kernels_affine(),      kernels_complex_cstr(), kernels_disease(), kernels_batch(),       kernels_photo(),
      kernels_cstr_series(), kernels_distillation(), kernels_polymer(), kernels_biofilm(),     kernels_heat_ex(),
      kernels_inv_batch(),   kernels_oscillators(),  kernels_user_stub(), kernels_me_sq(),     kernels_me_reactive_sq()};
  static_assert(PCG_MODEL_USER == 17 && PCG_KID_ME_SQ == 18 && PCG_KID_COUNT == 20, "kernel table order");
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
                                         DEF_HEAT_EX,     DEF_INV_BATCH,    DEF_OSCILLATORS, nullptr};

}  // namespace pcg

using namespace pcg;

struct pcg_plan {
  uint32_t magic;
  int device;
  int model_id, integrator_id;
  int kid;           // kernel-table id: model_id, or the *_SQ specialisation of the extraction models
  int lds_stages;
  int variant;       // PCG_OPT_VARIANT: 0 auto, 1 classic, 2 stream EPL=1, 3 stream EPL=2
  int stream_bpc;    // PCG_OPT_STREAM_BLOCKS_PER_CU: 0 = occupancy query
  int nt_stores;     // PCG_OPT_NT_STORES
  int num_cus;
  int stream_occ[2]; // resident workgroups per CU of the stream kernels [EPL-1] (0 = not queried yet)
  int pipe_occ[2][2];  // [auto-reset instantiation][EPL-1]
  int feat_occ[MAX_FEAT];  // resident workgroups per CU of the feature-masked kernels (0 = not queried yet)
  int q_bpc[2], q_tile[2]; // work-queue kernel [per_env_t]: resident workgroups per CU, tile slots (0 = not chosen yet,
                           // -1 = does not fit)
  int q_tile1[2];          // the largest tile with ONE workgroup per CU (Rodas4: launches that fit one tile per CU)
  int64_t env_offset;
  DevConst hc;       // host copy
  DevConst* dC;      // device copy
  double* dsched;    // [nsp+nd][N]
  size_t sched_bytes;
  LeanStep* dlean;   // [N] wave-uniform values of each lock-stepped step (inside the dsched allocation)
  int cfg_nu;        // na + ndm as the caller counts them
  hipFunction_t jit_fn[2];  // run-time compiled general step kernel with user expressions [per_env_t] (or null)
  hipFunction_t jit_integ, jit_rhs;  // PCG_MODEL_USER: the run-time compiled test hooks (pcg_integrate / pcg_rhs)
  hipFunction_t jit_roll;            // run-time compiled fused rollout of a plan with user expressions (or null)
  int nx;                   // states (the kernel table's for built-in models, the cfg's for PCG_MODEL_USER)
  int32_t* flat_ws;         // work space of the barrier-free rollout (pcg_rollout_flat.hpp): 4 counters + 2 x flat_cap indices,
  int64_t flat_cap;         // allocated at the first rollout that takes that path (and again if a later batch is larger)
};
static constexpr uint32_t PLAN_MAGIC = 0x50434731u;  // 'PCG1'

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return (int)_e;      \
  } while (0)

// Kernel-instantiation coverage (test infrastructure; off unless PCG_COVERAGE is set in the environment when the library is
// loaded).  Every launch site passes its kernel through cov(): with coverage on, the host function pointer (or, for a
// run-time compiled module, the hipFunction_t) is noted in a process-wide set.  pcg_coverage_names() turns the set into
// the kernels' mangled names -- the names tools/kernel_inventory.py reads out of this library's code objects -- so that
// the GPU suite can say which of the shipped instantiations it launched, test by test (tests/conftest.py).
static bool cov_on() {
  static const bool on = std::getenv("PCG_COVERAGE") != nullptr;
  return on;
}
static std::mutex g_cov_mu;
static std::set<const void*> g_cov_fn;        // ahead-of-time kernels: host function pointers
static std::set<hipFunction_t> g_cov_jit;     // run-time compiled kernels
template <class F>
static inline F cov(F fn) {
  if (cov_on()) {
    std::lock_guard<std::mutex> g(g_cov_mu);
    g_cov_fn.insert((const void*)fn);
  }
  return fn;
}
static inline hipFunction_t cov_jit(hipFunction_t fn) {
  if (cov_on()) {
    std::lock_guard<std::mutex> g(g_cov_mu);
    g_cov_jit.insert(fn);
  }
  return fn;
}

// Test hook: the work-queue kernel's tile sort on its own (the step results do not depend on the order, so no parity test
// can see a sort that does not sort -- only a slower launch would).
template <int E, int QB>
__global__ __launch_bounds__(QB) void sort_tile_test_kernel(uint32_t* w) {
  __shared__ uint32_t buf[E * QB];
  uint32_t* g = w + (size_t)blockIdx.x * E * QB;
  for (int i = threadIdx.x; i < E * QB; i += QB) buf[i] = g[i];
  __syncthreads();
  sort_tile<E, QB>(buf);
  for (int i = threadIdx.x; i < E * QB; i += QB) g[i] = buf[i];
}
extern "C" {

int pcg_version(void) { return PCG_ABI_VERSION; }

int64_t pcg_coverage_names(char* buf, int64_t cap, int reset) {
  std::string out;
  {
    std::lock_guard<std::mutex> g(g_cov_mu);
    for (const void* f : g_cov_fn) {
      const char* n = hipKernelNameRefByPtr(f, nullptr);
      out += n ? n : "?";
      out += '\n';
    }
    for (hipFunction_t f : g_cov_jit) {
      const char* n = hipKernelNameRef(f);
      out += "jit:";
      out += n ? n : "?";
      out += '\n';
    }
    if (reset) {
      g_cov_fn.clear();
      g_cov_jit.clear();
    }
  }
  if (buf && cap > 0) {
    const size_t n = std::min((size_t)cap - 1, out.size());
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return cov_on() ? (int64_t)out.size() + 1 : -1;
}

#ifndef PCG_SRC_HASH
#define PCG_SRC_HASH "unknown-build"  // the Makefile passes a digest of csrc/*.hpp, csrc/*.hip and include/pcgym_hip.h
#endif
const char* pcg_build_id(void) { return PCG_SRC_HASH; }


const char* pcg_strerror(int status) {
  switch (status) {
    case PCG_OK: return "ok";
    case PCG_E_NULL: return "required pointer is NULL";
    case PCG_E_MODEL: return "unknown model or integrator id";
    case PCG_E_DIM: return "dimension out of range or inconsistent";
    case PCG_E_VALUE: return "invalid scalar value";
    case PCG_E_PLAN: return "invalid plan handle or wrong device";
    case PCG_E_UNSUPPORTED: return "combination not supported by this build";
    case PCG_E_JIT: return "a user expression did not compile (see pcg_last_jit_log())";
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

int pcg_test_sort_tile(uint32_t* words, int32_t S, int32_t threads, int64_t ntiles, void* stream) {
  if (!words || ntiles <= 0 || ntiles > 0x7fffffff) return PCG_E_VALUE;
  const dim3 g((unsigned)ntiles);
  hipStream_t st = (hipStream_t)stream;
  if (threads == QBLOCK && S == QSORT / 4) hipLaunchKernelGGL(cov((sort_tile_test_kernel<QSORT / 4 / QBLOCK, QBLOCK>)), g, dim3(QBLOCK), 0, st, words);
  else if (threads == QBLOCK && S == QSORT / 2) hipLaunchKernelGGL(cov((sort_tile_test_kernel<QSORT / 2 / QBLOCK, QBLOCK>)), g, dim3(QBLOCK), 0, st, words);
  else if (threads == QBLOCK && S == QSORT) hipLaunchKernelGGL(cov((sort_tile_test_kernel<QSORT / QBLOCK, QBLOCK>)), g, dim3(QBLOCK), 0, st, words);
  else if (threads == 2 * QBLOCK && S == QSORT / 4) hipLaunchKernelGGL(cov((sort_tile_test_kernel<QSORT / 8 / QBLOCK, 2 * QBLOCK>)), g, dim3(2 * QBLOCK), 0, st, words);
  else if (threads == 2 * QBLOCK && S == QSORT / 2) hipLaunchKernelGGL(cov((sort_tile_test_kernel<QSORT / 4 / QBLOCK, 2 * QBLOCK>)), g, dim3(2 * QBLOCK), 0, st, words);
  else if (threads == 2 * QBLOCK && S == QSORT) hipLaunchKernelGGL(cov((sort_tile_test_kernel<QSORT / 2 / QBLOCK, 2 * QBLOCK>)), g, dim3(2 * QBLOCK), 0, st, words);
  else return PCG_E_UNSUPPORTED;
  return (int)hipGetLastError();
}

void pcg_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t o[4];
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], o);
  for (int i = 0; i < 4; ++i) out[i] = o[i];
}

// Validates cfg and fills the host DevConst.  No HIP calls: unit-testable without a GPU.
static int kernel_id_for(const pcg_env_cfg* c);
static int build_devconst(const pcg_env_cfg* c, DevConst* d, int* cfg_nu_out) {
  if (!c || !d) return PCG_E_NULL;
  if (c->model_id < 0 || c->model_id >= PCG_MODEL_COUNT) return PCG_E_MODEL;
  if (c->integrator_id < 0 || c->integrator_id >= PCG_INT_COUNT) return PCG_E_MODEL;
  const bool user = c->model_id == PCG_MODEL_USER;
  Kernels ku = kernels(c->model_id);
  if (user) {  // sizes come from the cfg; the right-hand side from cfg.user_rhs_src
    ku.nx = c->nx; ku.na = c->na; ku.ndm = c->ndm; ku.nraw = c->n_params;
  }
  const Kernels& k = ku;
  const int nx = c->nx, na = c->na, ndm = c->ndm, nd = c->nd, nsp = c->nsp, ncon = c->ncon, nrew = c->nrew;
  if (user) {
    if (!c->user_rhs_src) return PCG_E_NULL;
    if (nx < 1 || nx > PCG_MAX_NX || na < 1 || na > PCG_MAX_NA || ndm < 0 || ndm > PCG_MAX_NDM) return PCG_E_DIM;
    if (c->n_params < 0 || c->n_params > PCG_MAX_USER_PARAMS) return PCG_E_DIM;
    if (c->nunc > 0) return PCG_E_UNSUPPORTED;
  } else if (c->user_rhs_src) {
    return PCG_E_UNSUPPORTED;
  }
  if (user) {
  } else if (k.dynamic) {
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
  if (nunc > 0 && ndm > 0 && !c->d_param_index) return PCG_E_NULL;  // Q11: which parameter an unconfigured input reads
  if (nso != 0 && nso != nsp) return PCG_E_DIM;
  if (nd > 0 && nso != nsp) return PCG_E_UNSUPPORTED;
  if (nrew < 0 || nrew > PCG_MAX_NX) return PCG_E_DIM;
  if (c->N < 2 || c->N > PCG_MAX_N) return PCG_E_DIM;
  if (!(c->dt > 0.0) || !std::isfinite(c->dt)) return PCG_E_VALUE;
  if (c->integrator_id == PCG_INT_RK4 && c->substeps < 0) return PCG_E_VALUE;  // 0 = no integration (I/O probe)
  if ((c->integrator_id == PCG_INT_RK4G || c->integrator_id == PCG_INT_T5G) &&
      (c->substeps < 1 || !k.step[c->integrator_id][0][0][0] || c->nunc > 0))
    return c->substeps < 1 ? PCG_E_VALUE : PCG_E_UNSUPPORTED;  // models with a guard hook only
  if (c->integrator_id == PCG_INT_CV8 && (c->substeps < 1 || c->nunc > 0)) return c->substeps < 1 ? PCG_E_VALUE : PCG_E_UNSUPPORTED;
  if (c->integrator_id != PCG_INT_RK4 && (!(c->rtol > 0) || !(c->atol >= 0) || c->max_steps < 1))
    return PCG_E_VALUE;
  const int nobs = nx + nso + nd + nunc, cnu = na + ndm;
  if ((!c->params && !(user && c->n_params == 0)) || !c->x0 || !c->a_low || !c->a_high || !c->o_low || !c->o_high) return PCG_E_NULL;
  if (nsp && (!c->sp_index || !c->sp)) return PCG_E_NULL;
  if ((nsp || nrew) && !c->r_scale) return PCG_E_NULL;
  if (nrew && !c->rew_index) return PCG_E_NULL;
  if (nd && (!c->d_slot || !c->d_sched)) return PCG_E_NULL;
  if (ndm && !c->d_default) return PCG_E_NULL;
  if (ncon && !c->user_cons_src && (!c->con_A || !c->con_b)) return PCG_E_NULL;
  if ((c->user_cons_src || c->user_reward_src) && (k.dynamic || c->nunc > 0)) return PCG_E_UNSUPPORTED;
  if ((c->flags & PCG_F_A_DELTA) && (!c->a_act_low || !c->a_act_high || !c->a_0)) return PCG_E_NULL;
  if ((c->flags & PCG_F_NOISE) && !c->noise_pct) return PCG_E_NULL;
  if ((c->flags & PCG_F_GAUSS_DIST) && nd && (!c->d_sigma || !c->d_clip_lo || !c->d_clip_hi)) return PCG_E_NULL;

  std::memset(d, 0, sizeof(*d));
  double ddef[PCG_MAX_NDM] = {0, 0, 0, 0};
  if (user) {
    static_assert(PCG_MAX_USER_PARAMS <= sizeof(d->kp_big) / sizeof(double), "user parameters live in kp_big");
    for (int i = 0; i < c->n_params; ++i) d->kp_big[i] = c->params[i];
  } else {
    k.prep(c->params, nx, na, k.dynamic ? d->kp_big : d->kp, ddef);
  }
  for (int j = 0; j < k.ndm; ++j) d->d_default[j] = ndm ? c->d_default[j] : ddef[j];
  const bool norm_a = c->flags & PCG_F_NORMALISE_A, norm_o = c->flags & PCG_F_NORMALISE_O;
  const bool compat = c->flags & PCG_F_REF_COMPAT;
  for (int i = 0; i < na; ++i) {
    d->a_lo[i] = c->a_low[i];
    d->a_hi[i] = c->a_high[i];
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
  if (c->flags & PCG_F_REWARD_TRACK) {
    // normalisation of the tracking reward is always by o_space / a_space (custom_reward.py:14-31), whether or
    // not the env normalises its observations / actions
    if ((c->flags & PCG_F_REWARD_BATCH) || nsp == 0) return PCG_E_UNSUPPORTED;
    if ((c->flags & PCG_F_REWARD_CRYST) && c->model_id != PCG_MODEL_CRYST) return PCG_E_UNSUPPORTED;
    if (c->rew_nbox < 0 || c->rew_nbox > PCG_MAX_RBOX) return PCG_E_DIM;
    if (c->rew_nbox > 0 && (!c->rew_box_index || !c->rew_box_lo || !c->rew_box_hi)) return PCG_E_NULL;
    if (nd > 0) return PCG_E_UNSUPPORTED;  // the reference broadcasts uk (Nu + Nd) against a_space (Nu) there
    for (int k = 0; k < nsp; ++k) {
      const int i = c->sp_index[k];
      const double w = c->o_high[i] - c->o_low[i];
      if (!(w != 0.0)) return PCG_E_VALUE;
      d->trk_lo[k] = c->o_low[i];
      d->trk_inv[k] = 1.0 / w;
    }
    for (int j = 0; j < na; ++j) {
      const double w = c->a_high[j] - c->a_low[j];
      if (!(w != 0.0)) return PCG_E_VALUE;
      d->act_lo[j] = c->a_low[j];
      d->act_inv[j] = 1.0 / w;
    }
    d->R_du = c->rew_R_du;
    d->R_u = c->rew_R_u;
    d->nbox = c->rew_nbox;
    for (int q = 0; q < c->rew_nbox; ++q) {
      const int i = c->rew_box_index[q];
      if (i < 0 || i >= nx) return PCG_E_DIM;
      const double w = c->o_high[i] - c->o_low[i];
      if (!(w != 0.0)) return PCG_E_VALUE;
      d->box_index[q] = i;
      d->box_lo[q] = c->o_low[i];
      d->box_inv[q] = 1.0 / w;
      d->box_lon[q] = (c->rew_box_lo[q] - c->o_low[i]) / w;
      d->box_hin[q] = (c->rew_box_hi[q] - c->o_low[i]) / w;
    }
  }
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
  // the same quirk for user constraint expressions, as an affine map of the vectors they index
  for (int i = 0; i < PCG_MAX_NOBS; ++i) { d->q3_mul[i] = 1.0; d->q3_add[i] = 0.0; }
  for (int j = 0; j < KNU; ++j) { d->q3u_mul[j] = 1.0; d->q3u_add[j] = 0.0; }
  if (compat && norm_o)
    for (int i = 0; i < nobs; ++i) {
      const double hs = (c->o_high[i] - c->o_low[i]) / 2;
      d->q3_mul[i] = hs;
      d->q3_add[i] = hs + c->o_low[i];
    }
  if (compat && norm_a && c->user_cons_src) {
    if (cnu != na && na != 1) return PCG_E_UNSUPPORTED;  // the reference itself raises (broadcast error)
    for (int j = 0; j < cnu; ++j) {
      const int q = (na == 1) ? 0 : j;
      const double hs = (c->a_high[q] - c->a_low[q]) / 2;
      const int col = (j < na) ? j : k.na + (j - na);  // kernel-side u layout [NA | NDM]
      d->q3u_mul[col] = hs;
      d->q3u_add[col] = hs + c->a_low[q];
    }
  }
  for (int r = 0; r < (c->user_cons_src ? 0 : ncon); ++r) {
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
  if (!k.dynamic && !user)
    for (int i = 0; i < k.nraw && i < 32; ++i) d->raw[i] = c->params[i];
  for (int j = 0; j < nunc; ++j) {
    // index == nraw: an INERT entry (sampled at reset and observed, substituted into no model parameter) -- what the
    // reference does with empirical_distribution['x0'] (pcgym.py:311-316); empirical tables only
    const bool inert = (c->flags & PCG_F_UNC_EMPIRICAL) && c->unc_index[j] == k.nraw && k.nraw < 32;
    if ((c->unc_index[j] < 0 || c->unc_index[j] >= k.nraw) && !inert) return PCG_E_DIM;
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
  d->h2 = 0.5 * d->h;
  d->h6 = d->h / 6.0;
  d->rtol = c->rtol;
  d->dt_edge = c->dt * (1.0 - 1e-14);
  d->h_floor = 1e-13 * c->dt;
  // end-point error control of the Rosenbrock pairs (PCG_INT_RODAS4 / PCG_INT_RODAS5, pcgym_hip.h): exponent rate = ep_c x the
  // model's contraction rate
  if (is_ros_pair(c->integrator_id)) {
    if (!(c->ep_frac >= 0.0 && c->ep_frac <= 1.0) || c->ep_kmax < 0 || c->ep_kmax > 40) return PCG_E_VALUE;
    if (c->nunc > 0) return PCG_E_UNSUPPORTED;
    d->ep_kmax = c->ep_kmax;
    d->ep_c = c->ep_kmax > 0 ? c->ep_frac * 1.4426950408889634 : 0.0;
  }
  // cooperative rule (pcgym_hip.h: coop_thr): only where the model's kernels carry it
  d->coop_thr = 0.0;
  if (c->coop_thr != 0.0) {
    if (!(c->coop_thr > 0.0) || !std::isfinite(c->coop_thr)) return PCG_E_VALUE;
    if (!is_ros_pair(c->integrator_id) || user || !c->params || !kernels(kernel_id_for(c)).coop) return PCG_E_UNSUPPORTED;
    d->coop_thr = c->coop_thr;
  }
  d->atol = c->atol;
  d->nx = nx; d->na = na; d->ndm = ndm; d->nd = nd; d->nsp = nsp; d->nsp_obs = nso; d->ncon = ncon; d->nrew = nrew;
  d->N = c->N; d->substeps = c->substeps; d->max_steps = c->max_steps; d->nobs = nobs;
  d->flags = c->flags;
  if (cfg_nu_out) *cfg_nu_out = cnu;
  return PCG_OK;
}

// The wave-uniform values of every lock-stepped step t -> t + 1 (LeanStep, pcg_kernels.hpp), computed here once per plan
// with the operations the kernels used to repeat per tile: the observation slots through the folded OMap as one fused
// multiply-add of (v - lo), the schedule indices of pcgym.py:394,438,555 (quirks Q5, Q6) clamped to the last column.
static std::vector<LeanStep> lean_table(const DevConst& d, const pcg_env_cfg* c) {
  const int N = d.N > 0 ? d.N : 1;
  std::vector<LeanStep> tab((size_t)N);
  for (int t = 0; t < N; ++t) {
    LeanStep& L = tab[(size_t)t];
    std::memset(&L, 0, sizeof(L));
    const int tc = t < N - 1 ? t : N - 1, tn = t + 1 < N - 1 ? t + 1 : N - 1;
    for (int k = 0; k < d.nsp && k < PCG_MAX_NSP; ++k) {
      L.spn[k] = c->sp[(size_t)k * N + tn];
      if (k < d.nsp_obs) {
        const OMap& m = d.omap[d.nx + k];
        L.osp[k] = std::fma(c->sp[(size_t)k * N + tc] - m.lo, m.sc, m.off);
      }
    }
    for (int j = 0; j < PCG_MAX_NDM; ++j) L.ud[j] = d.d_default[j];
    for (int k = 0; k < d.nd && k < PCG_MAX_NDM; ++k) {
      const double v = c->d_sched[(size_t)k * N + tn];
      const OMap& m = d.omap[d.nx + d.nsp_obs + k];
      L.od[k] = std::fma(v - m.lo, m.sc, m.off);
      const int slot = d.d_slot[k];
      if (slot >= 0 && slot < PCG_MAX_NDM) L.ud[slot] = v;
    }
  }
  return tab;
}

// eq_exponent == 2 (the reference default) selects the multiply-only instantiation of the extraction models;
// not when the exponent itself is one of the per-env uncertain parameters.
static int kernel_id_for(const pcg_env_cfg* c) {
  int epos = -1, kid = c->model_id;
  if (c->model_id == PCG_MODEL_ME) { epos = 4; kid = PCG_KID_ME_SQ; }
  if (c->model_id == PCG_MODEL_ME_REACTIVE) { epos = 5; kid = PCG_KID_ME_REACTIVE_SQ; }
  if (epos < 0 || c->params[epos] != 2.0) return c->model_id;
  for (int j = 0; j < c->nunc; ++j)
    if (c->unc_index[j] == epos) return c->model_id;
  return kid;
}

// ---- run-time compilation of user source (pcgym_hip.h: user_cons_src / user_reward_src / user_rhs_src) --------------
// The general one-env-per-lane step kernel of the plan's model is instantiated from the library's own headers with the
// user's source spliced in as pcg_user_constraints / pcg_user_reward / pcg_user_rhs, compiled with hipRTC, loaded as
// a module.  A PCG_MODEL_USER plan has no ahead-of-time kernels at all: its module also carries the test hooks.
struct JitModule {
  hipFunction_t fn[2];      // step_kernel [per_env_t]
  hipFunction_t integ, rhs; // PCG_MODEL_USER only
  hipFunction_t roll;       // fused rollout kernel (null for the Rosenbrock integrators: their matrices live in LDS)
};
static std::mutex g_jit_mu;
static std::map<uint64_t, JitModule> g_jit_cache;
static std::string g_jit_log;

static uint64_t fnv1a(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char ch : s) h = (h ^ ch) * 1099511628211ull;
  return h;
}

static uint64_t fnv1a_seed(const std::string& s, uint64_t seed) {
  uint64_t h = 1469598103934665603ull ^ seed;
  for (unsigned char ch : s) h = (h ^ ch) * 1099511628211ull;
  return h;
}
#ifndef PCG_SRC_HASH
#define PCG_SRC_HASH "unknown-build"  // the Makefile passes a digest of csrc/*.hpp, csrc/*.hip and include/pcgym_hip.h
#endif

// digest of the headers a run-time compilation will see: every *.hpp of the include directory and the ABI header
// next to it (../../include/pcgym_hip.h), by content.  Cached per directory for the life of the process.
static std::string jit_header_digest(const char* inc_dir) {
  static std::map<std::string, std::string> memo;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  auto it = memo.find(inc_dir);
  if (it != memo.end()) return it->second;
  std::vector<std::string> files;
  if (DIR* d = ::opendir(inc_dir)) {
    while (dirent* e = ::readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() > 4 && n.compare(n.size() - 4, 4, ".hpp") == 0) files.push_back(std::string(inc_dir) + "/" + n);
    }
    ::closedir(d);
  }
  std::sort(files.begin(), files.end());
  files.push_back(std::string(inc_dir) + "/../../include/pcgym_hip.h");
  uint64_t h1 = 0, h2 = 0;
  for (const std::string& f : files) {
    std::ifstream in(f, std::ios::binary);
    const std::string body((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const std::string tag = f.substr(f.find_last_of('/') + 1) + ":" + std::to_string(body.size()) + ":";
    h1 = fnv1a_seed(tag + body, h1);
    h2 = fnv1a_seed(tag + body, h2 ^ 0x9E3779B97F4A7C15ull);
  }
  char hex[40];
  std::snprintf(hex, sizeof(hex), "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
  return memo[inc_dir] = hex;
}

// The disk cache lives in a directory only its owner can write: $PCG_JIT_CACHE, else $XDG_CACHE_HOME/pcgym_amd, else
// ~/.cache/pcgym_amd -- created with mkdir(2), mode 0700, no shell.  A directory (or file) that belongs to somebody
// else, or that group / others can write, is not used: code objects found there would run inside this process.
// Returns "" when there is no usable directory (the in-process cache still works).
static bool jit_path_private(const std::string& p, bool want_dir) {
  struct stat st;
  if (::lstat(p.c_str(), &st) != 0) return false;
  if (want_dir ? !S_ISDIR(st.st_mode) : !S_ISREG(st.st_mode)) return false;
  return st.st_uid == ::geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
static std::string jit_cache_dir() {
  std::string dir;
  if (const char* e = std::getenv("PCG_JIT_CACHE")) dir = e;
  else if (const char* x = std::getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/pcgym_amd";
  else if (const char* h = std::getenv("HOME")) dir = std::string(h) + "/.cache/pcgym_amd";
  if (dir.empty() || dir[0] != '/') return std::string();
  for (size_t i = 1; i <= dir.size(); ++i)  // mkdir -p, without a shell
    if (i == dir.size() || dir[i] == '/') {
      const std::string part = dir.substr(0, i);
      if (::mkdir(part.c_str(), 0700) != 0 && errno != EEXIST) return std::string();
    }
  return jit_path_private(dir, true) ? dir : std::string();
}
// file = "PCGJIT2\n" nfn lowered names (one per line) code size, two 64-bit digests of (names + code) "\n" code
static bool jit_cache_read(const std::string& path, int nfn, std::string* code, std::string* low) {
  if (!jit_path_private(path, false)) return false;
  std::ifstream in(path, std::ios::binary);
  std::string magic, line;
  if (!std::getline(in, magic) || magic != "PCGJIT2") return false;
  std::string all;
  for (int q = 0; q < nfn; ++q) {
    if (!std::getline(in, low[q]) || low[q].empty()) return false;
    all += low[q] + "\n";
  }
  unsigned long long sz = 0, d1 = 0, d2 = 0;
  if (!std::getline(in, line) || std::sscanf(line.c_str(), "%llu %llx %llx", &sz, &d1, &d2) != 3 || sz == 0 || sz > (1ull << 30))
    return false;
  std::string body((size_t)sz, '\0');
  in.read(&body[0], (std::streamsize)sz);
  if ((unsigned long long)in.gcount() != sz) return false;
  all += body;
  if (fnv1a(all) != d1 || fnv1a_seed(all, 0x9E3779B97F4A7C15ull) != d2) return false;  // truncated / corrupted / edited
  *code = std::move(body);
  return true;
}
static void jit_cache_write(const std::string& path, int nfn, const std::string& code, const std::string* low) {
  // several processes (one per GPU) may compile the same source at once: a private temporary, complete and closed
  // before it appears under the final name; a write error never publishes a file
  const std::string tmp = path + "." + std::to_string((long long)getpid()) + ".tmp";
  const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
  if (fd < 0) return;
  std::string all;
  for (int q = 0; q < nfn; ++q) all += low[q] + "\n";
  std::string head = "PCGJIT2\n" + all;
  all += code;
  char meta[96];
  std::snprintf(meta, sizeof(meta), "%llu %016llx %016llx\n", (unsigned long long)code.size(), (unsigned long long)fnv1a(all),
                (unsigned long long)fnv1a_seed(all, 0x9E3779B97F4A7C15ull));
  head += meta;
  bool ok = true;
  const std::string* parts[2] = {&head, &code};
  for (const std::string* part : parts) {
    size_t off = 0;
    while (ok && off < part->size()) {
      const ssize_t w = ::write(fd, part->data() + off, part->size() - off);
      if (w <= 0) ok = false;
      else off += (size_t)w;
    }
  }
  ok = (::close(fd) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) ::unlink(tmp.c_str());
}

static int jit_kernels(const pcg_env_cfg* cfg, int kid, int device, JitModule* out) {
  if (!cfg->jit_include_dir) return PCG_E_NULL;
  const bool user = cfg->model_id == PCG_MODEL_USER;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  std::string arch = prop.gcnArchName;
  arch = arch.substr(0, arch.find(':'));
  std::ostringstream src;
  if (user)
    src << "#define PCG_USER_NX " << cfg->nx << "\n#define PCG_USER_NA " << cfg->na << "\n#define PCG_USER_NDM " << cfg->ndm
        << "\n#define PCG_USER_NP " << cfg->n_params << "\n";
  if (cfg->user_cons_src) src << "#define PCG_USER_NCON " << cfg->ncon << "\n";
  if (cfg->user_reward_src) src << "#define PCG_USER_REWARD 1\n";
  src << "#include \"pcg_kernels.hpp\"\n"
      << "static_assert(sizeof(pcg::DevConst) == " << sizeof(DevConst) << " && sizeof(pcg::StepArgs) == " << sizeof(StepArgs)
      << ", \"kernel headers differ from the ones libpcgym_hip.so was built from\");\n"
      << "namespace pcg {\n";
  if (user)
    src << "__device__ void pcg_user_rhs(const double* x, const double* u, const double* p, double* dx) {\n"
        << cfg->user_rhs_src << "\n}\n";
  if (cfg->user_cons_src)
    src << "__device__ void pcg_user_constraints(const double* x, const double* u, double* g) {\n" << cfg->user_cons_src
        << "\n}\n";
  if (cfg->user_reward_src)
    src << "__device__ double pcg_user_reward(const double* o, const double* x, const double* u, const double* sp, "
           "int violated, int t, int N) {\n  return (double)(" << cfg->user_reward_src << ");\n}\n";
  src << "}\n";
  const bool roll = cfg->integrator_id != PCG_INT_RODAS3 && !is_ros_pair(cfg->integrator_id);
  const int iroll = user ? 4 : 2;
  const int nfn = iroll + (roll ? 1 : 0);
  std::string names[5];
  for (int pe = 0; pe < 2; ++pe) {
    std::ostringstream nm;
    nm << "pcg::step_kernel<pcg::Model<" << kid << ">, " << cfg->integrator_id << ", " << (pe ? "true" : "false")
       << ", false, true, false>";
    names[pe] = nm.str();
    src << "template __global__ void " << names[pe] << "(const pcg::StepArgs);\n";
  }
  if (user) {
    std::ostringstream ni, nr;
    ni << "pcg::integrate_kernel<pcg::Model<" << kid << ">, " << cfg->integrator_id << ", false>";
    nr << "pcg::rhs_kernel<pcg::Model<" << kid << "> >";
    names[2] = ni.str();
    names[3] = nr.str();
    src << "template __global__ void " << names[2] << "(pcg::CDevConst*, int64_t, int, double*, const double*, int32_t*);\n";
    src << "template __global__ void " << names[3] << "(pcg::CDevConst*, int64_t, int, const double*, const double*, double*);\n";
  }
  if (roll) {  // pcg_rollout for plans with user expressions: T steps with the state in registers
    std::ostringstream nm;
    nm << "pcg::rollout_kernel<pcg::Model<" << kid << ">, " << cfg->integrator_id << ", false>";
    names[iroll] = nm.str();
    src << "template __global__ void " << names[iroll] << "(const pcg::StepArgs);\n";
  }
  const std::string text = src.str();
  // The translation unit only says `#include "pcg_kernels.hpp"`: the CONTENT of the kernel headers it will be compiled
  // against has to be part of the key too (a changed integrator or model with unchanged struct sizes must not find an
  // old code object) -- hash of every header in the include directory + the ABI header + the library's own build id.
  const std::string hdr = jit_header_digest(cfg->jit_include_dir);
  const std::string ident = text + "|" + arch + "|" + std::to_string(PCG_ABI_VERSION) + "|" + hdr + "|" PCG_SRC_HASH;
  const uint64_t key = fnv1a(ident + "|" + std::to_string(device));
  std::lock_guard<std::mutex> lk(g_jit_mu);
  auto hit = g_jit_cache.find(key);
  if (hit != g_jit_cache.end()) {
    *out = hit->second;
    return PCG_OK;
  }
  // disk cache: one file per source = lowered kernel names + code object, with a digest of both
  const std::string dir = jit_cache_dir();
  char hex[40];
  std::snprintf(hex, sizeof(hex), "%016llx%016llx", (unsigned long long)fnv1a(ident), (unsigned long long)fnv1a_seed(ident, 0x9E3779B97F4A7C15ull));
  const std::string path = dir.empty() ? std::string() : dir + "/" + hex + ".pco";
  std::string code, low[5];
  bool from_disk = !path.empty() && jit_cache_read(path, nfn, &code, low);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (code.empty()) {
      hiprtcProgram prog;
      if (hiprtcCreateProgram(&prog, text.c_str(), "pcg_user_step.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return PCG_E_JIT;
      for (int q = 0; q < nfn; ++q) hiprtcAddNameExpression(prog, names[q].c_str());
      const std::string oarch = "--offload-arch=" + arch, oinc = std::string("-I") + cfg->jit_include_dir;
      const char* opts[] = {oarch.c_str(), "-O3", "-std=c++17", oinc.c_str()};
      const hiprtcResult cr = hiprtcCompileProgram(prog, 4, opts);
      size_t ls = 0;
      hiprtcGetProgramLogSize(prog, &ls);
      g_jit_log.assign(ls, '\0');
      if (ls) hiprtcGetProgramLog(prog, &g_jit_log[0]);
      if (cr != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        return PCG_E_JIT;
      }
      for (int q = 0; q < nfn; ++q) {
        const char* ln = nullptr;
        if (hiprtcGetLoweredName(prog, names[q].c_str(), &ln) != HIPRTC_SUCCESS || !ln) {
          hiprtcDestroyProgram(&prog);
          return PCG_E_JIT;
        }
        low[q] = ln;
      }
      size_t cs = 0;
      hiprtcGetCodeSize(prog, &cs);
      code.assign(cs, '\0');
      hiprtcGetCode(prog, &code[0]);
      hiprtcDestroyProgram(&prog);
      if (!path.empty()) jit_cache_write(path, nfn, code, low);  // best effort: an unwritable cache is only slower next time
    }
    hipModule_t mod;
    const hipError_t le = hipModuleLoadData(&mod, code.data());
    if (le != hipSuccess) {
      if (from_disk && attempt == 0) {  // a cached object the driver refuses: drop it and compile afresh
        (void)hipGetLastError();
        ::unlink(path.c_str());
        code.clear();
        from_disk = false;
        continue;
      }
      return (int)le;
    }
    JitModule jm;
    jm.integ = jm.rhs = jm.roll = nullptr;
    if (roll) HIP_TRY(hipModuleGetFunction(&jm.roll, mod, low[iroll].c_str()));
    for (int pe = 0; pe < 2; ++pe) HIP_TRY(hipModuleGetFunction(&jm.fn[pe], mod, low[pe].c_str()));
    if (user) {
      HIP_TRY(hipModuleGetFunction(&jm.integ, mod, low[2].c_str()));
      HIP_TRY(hipModuleGetFunction(&jm.rhs, mod, low[3].c_str()));
    }
    g_jit_cache[key] = jm;
    *out = jm;
    return PCG_OK;
  }
  return PCG_E_JIT;
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
  p->kid = kernel_id_for(cfg);
  p->integrator_id = cfg->integrator_id;
  p->lds_stages = 0;
  p->variant = 0;
  p->stream_bpc = 0;
  p->nt_stores = 1;  // measured: 20.3 -> 18.7 us per launch on the cstr workload (profiles/)
  p->num_cus = 0;
  p->stream_occ[0] = p->stream_occ[1] = 0;
  p->pipe_occ[0][0] = p->pipe_occ[0][1] = p->pipe_occ[1][0] = p->pipe_occ[1][1] = 0;
  for (int i = 0; i < MAX_FEAT; ++i) p->feat_occ[i] = 0;
  p->q_bpc[0] = p->q_bpc[1] = p->q_tile[0] = p->q_tile[1] = p->q_tile1[0] = p->q_tile1[1] = 0;
  p->env_offset = 0;
  p->dC = nullptr;
  p->dsched = nullptr;
  p->dlean = nullptr;
  p->jit_fn[0] = p->jit_fn[1] = nullptr;
  p->jit_integ = p->jit_rhs = p->jit_roll = nullptr;
  p->nx = cfg->nx;
  hipError_t e = hipGetDevice(&p->device);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&p->num_cus, hipDeviceAttributeMultiprocessorCount, p->device);
  if (e != hipSuccess) { delete p; return (int)e; }
  const int rows = cfg->nsp + cfg->nd;
  const bool emp = (cfg->flags & PCG_F_UNC_EMPIRICAL) && cfg->nunc > 0;
  const size_t n_emp = emp ? (size_t)cfg->unc_emp_off[cfg->nunc] : 0;
  p->sched_bytes = sizeof(double) * (size_t)(rows > 0 ? rows : 1) * cfg->N;
  // behind the schedule rows and the sample tables: the per-step table of the lean kernels (LeanStep[N], 128-byte records)
  const size_t lean_off = (((size_t)(rows > 0 ? rows : 1) * cfg->N + n_emp) + 15) & ~(size_t)15;  // in doubles
  const size_t sched_alloc = sizeof(double) * lean_off + sizeof(LeanStep) * (size_t)cfg->N;
  e = hipMalloc((void**)&p->dC, sizeof(DevConst));
  if (e == hipSuccess) e = hipMalloc((void**)&p->dsched, sched_alloc);
  if (e == hipSuccess) {
    p->dlean = reinterpret_cast<LeanStep*>(p->dsched + lean_off);
    const std::vector<LeanStep> lt = lean_table(p->hc, cfg);
    e = hipMemcpy(p->dlean, lt.data(), sizeof(LeanStep) * lt.size(), hipMemcpyHostToDevice);
  }
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
  if (cfg->user_cons_src || cfg->user_reward_src || cfg->user_rhs_src) {
    JitModule jm;
    rc = jit_kernels(cfg, p->kid, p->device, &jm);
    if (rc != PCG_OK) {
      (void)hipFree(p->dC);
      (void)hipFree(p->dsched);
      delete p;
      return rc;
    }
    p->jit_fn[0] = jm.fn[0];
    p->jit_fn[1] = jm.fn[1];
    p->jit_integ = jm.integ;
    p->jit_rhs = jm.rhs;
    p->jit_roll = jm.roll;
    // Rosenbrock pairs keep nx^2 doubles per lane in LDS: past 48 KB per workgroup (nx >= 10) a kernel has to be told.
    // Decided HERE, so that a plan that cannot run says so at creation and not at its first step.
    const int jnx = cfg->model_id == PCG_MODEL_USER ? cfg->nx : kernels(p->kid).nx;
    const size_t need = sizeof(double) * integ_lds_doubles(jnx, cfg->integrator_id, false) +
                        sizeof(double) * (size_t)(p->hc.nsp + p->hc.nd) * p->hc.N;
    if (need > 48 * 1024) {
      hipFunction_t fns[3] = {jm.fn[0], jm.fn[1], jm.integ};
      for (hipFunction_t f : fns) {
        if (!f) continue;
        const hipError_t ae = hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)(need < 160 * 1024 ? need : 160 * 1024));
        if (ae != hipSuccess) {
          (void)hipGetLastError();
          (void)hipFree(p->dC);
          (void)hipFree(p->dsched);
          delete p;
          return PCG_E_UNSUPPORTED;
        }
      }
    }
  }
  *out = p;
  return PCG_OK;
}

const char* pcg_last_jit_log(void) { return g_jit_log.c_str(); }

static bool plan_ok(const pcg_plan* p) { return p && p->magic == PLAN_MAGIC; }

int pcg_plan_destroy(pcg_plan* p) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  p->magic = 0;
  hipError_t e1 = hipFree(p->dC), e2 = hipFree(p->dsched);
  if (p->flat_ws) (void)hipFree(p->flat_ws);
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
    case PCG_OPT_NT_STORES: p->nt_stores = (int)(value & 7); return PCG_OK;  // bit 0 obs / reward, 1 state, 2 loads
    case PCG_OPT_VARIANT:
      if (value < 0 || value > 5) return PCG_E_VALUE;
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
  if ((c.flags & PCG_F_REWARD_TRACK) && io->u_prev) A += 16 * c.na;
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
  a->lean = (const PCG_CONSTANT LeanStep*)p->dlean;
  a->x = io->x; a->a = io->a; a->d = io->d; a->t = io->t; a->a_save = io->a_save; a->obs = io->obs;
  a->rew = io->rew; a->done = io->done; a->viol = io->viol; a->g = io->g; a->g_pre = io->g_pre;
  a->nsteps = io->nsteps; a->B = io->B; a->env_offset = p->env_offset;
  a->p_unc = io->p_unc;
  a->u_prev = io->u_prev;
  a->status = io->status;
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

// Launch geometry of the DOPRI5 work-queue kernel (pcg_step_queue.hpp): resident 256-thread workgroups per CU by the
// register allocation (= waves per SIMD), the largest tile (<= 1024 slots, four per lane) whose LDS fits that many
// workgroups in 160 KB.
static int queue_geometry(pcg_plan* p, const Kernels& k, int pe, size_t sched_bytes) {
  if (p->q_tile[pe] != 0) return PCG_OK;
  hipFuncAttributes fa;
  const bool guarded = p->integrator_id == PCG_INT_RK4G || p->integrator_id == PCG_INT_T5G;  // their fix-up launch
  const bool ros = is_ros_pair(p->integrator_id);
  const StepFn* q_w1 = p->integrator_id == PCG_INT_RODAS5 ? k.queue_r5w1 : k.queue_r4w1;
  const StepFn qfn = (p->integrator_id == PCG_INT_RODAS5 ? k.queue_r5 : ros ? k.queue_r4 : guarded ? k.queue_fix : k.queue)[pe];
  hipError_t e = hipFuncGetAttributes(&fa, (const void*)qfn);
  if (e != hipSuccess) return (int)e;
  const int alloc = ((fa.numRegs + 7) / 8) * 8;
  int bpc = alloc > 0 ? 512 / alloc : 1;
  bpc = bpc < 1 ? 1 : (bpc > 4 ? 4 : bpc);
  if (const char* ev = std::getenv("PCG_Q_BPC")) {  // measurement switch: fewer resident workgroups per CU
    const int v = std::atoi(ev);
    if (v >= 1 && v < bpc) bpc = v;
  }
  const size_t lds_cu = 160 * 1024 - 2048;
  // tile cap: four envs per lane for the explicit pair (tuned in round 2); the Rosenbrock pair's attempts per env are
  // heavy-tailed (median 17, 1 % above 70, maximum ~100 on BASELINE configs[2]) and want the largest pool
  const int tcap = ros ? QSORT : QSORT / 2;
  int best_t = 0, best_b = 1, t1 = 0;
  for (int b = bpc; b >= 1; --b) {
    int T = tcap;
    while (T >= QBLOCK && k.queue_lds(T) + sched_bytes > lds_cu / b) T -= 64;
    if (T < QBLOCK) continue;
    if (b == 1) t1 = T;
    if (b * T > best_b * best_t) {
      best_t = T;
      best_b = b;
    }
    if (T == tcap && !ros) break;  // the full tile at the highest occupancy that allows it
  }
  p->q_tile[pe] = best_t > 0 ? best_t : -1;
  p->q_bpc[pe] = best_b;
  p->q_tile1[pe] = t1;
  if (best_t > 0) {
    // (the whole CU's LDS: a launch may also park its tile's state there when that fits, see step_impl)
    e = hipFuncSetAttribute((const void*)qfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu);
    if (e != hipSuccess) return (int)e;
    if (ros && q_w1[pe]) {
      e = hipFuncSetAttribute((const void*)q_w1[pe], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu);
      if (e != hipSuccess) return (int)e;
    }
    if (p->integrator_id == PCG_INT_DOPRI5 && k.queue_w[pe]) {
      e = hipFuncSetAttribute((const void*)k.queue_w[pe], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu);
      if (e != hipSuccess) return (int)e;
    }
  }
  return PCG_OK;
}

// Fill every lazily queried occupancy of the plan's candidate persistent kernels (done before a stream
// capture so that no query runs while capturing).
static int warm_occupancy(pcg_plan* p) {
  const Kernels& k = kernels(p->kid);
  for (int e = 0; e < 2; ++e) {
    const int ls = lean_scheme(p->integrator_id);
    if (ls >= 0 && k.pipe[ls][e] && p->pipe_occ[0][e] == 0) {
      const int q = resident_blocks(k.pipe[ls][e]);
      if (q < 0) return -q;
      p->pipe_occ[0][e] = q;
    }
    if (k.stream[p->integrator_id][e] && p->stream_occ[e] == 0) {
      const int q = resident_blocks(k.stream[p->integrator_id][e]);
      if (q < 0) return -q;
      p->stream_occ[e] = q;
    }
  }
  if ((p->integrator_id == PCG_INT_DOPRI5 && k.queue[0]) || ((p->integrator_id == PCG_INT_RK4G || p->integrator_id == PCG_INT_T5G) && k.queue_fix[0]) ||
      (is_ros_pair(p->integrator_id) && k.queue_r4[0])) {
    const int rc = queue_geometry(p, k, 0, 0);
    if (rc != PCG_OK) return rc;
  }
  if (p->integrator_id == PCG_INT_RK4)
    for (int i = 0; i < k.nfeat; ++i)
      if (p->feat_occ[i] == 0) {
        const int q = resident_blocks(k.feat[i].fn);
        if (q < 0) return -q;
        p->feat_occ[i] = q;
      }
  return PCG_OK;
}

static int step_impl(pcg_plan* p, const pcg_buffers* io, int32_t t, uint64_t seed, void* stream, bool auto_reset,
                     uint64_t reset_seed) {
  StepArgs a;
  int rc = fill_args(p, io, &a);
  a.auto_reset = auto_reset ? 1 : 0;
  a.reset_seed = reset_seed;
  if (rc != PCG_OK) return rc;
  if (io->B == 0) return PCG_OK;  // empty batch: nothing to do (zero-size buffers may be NULL)
  if (!io->x || !io->a || !io->obs || !io->rew || !io->done) return PCG_E_NULL;
  const DevConst& c = p->hc;
  if ((c.flags & PCG_F_A_DELTA) && !io->a_save) return PCG_E_NULL;
  if ((c.flags & PCG_F_REWARD_TRACK) && !io->u_prev) return PCG_E_NULL;
  a.t_scalar = t;
  a.seed = seed;
  const bool per_env_t = io->t != nullptr;
  if (!per_env_t && t < 0) return PCG_E_VALUE;  // the lock-stepped counter indexes the schedules (t past N-1 clamps, t < 0 cannot)
  const Kernels& k = kernels(p->kid);
  const bool lds_st = p->lds_stages && p->integrator_id == PCG_INT_DOPRI5 && k.has_lds_stages;
  const int knx = p->model_id == PCG_MODEL_USER ? p->nx : k.nx;  // the kernels' compile-time state count
  const bool rstr = p->model_id != PCG_MODEL_USER && k.ros_structured;  // Rodas4 with the model's own W: no LDS
  const int block = tb(lds_st, p->integrator_id, knx, rstr);
  size_t shmem = sizeof(double) * integ_lds_doubles(knx, p->integrator_id, lds_st, rstr);
  const size_t integ_shmem = shmem;
  if (per_env_t) {
    const size_t sb = sizeof(double) * (size_t)(c.nsp + c.nd) * c.N;
    if (sb > 0 && shmem + sb <= (shmem > 48 * 1024 ? 160 : 64) * 1024) {
      a.sched_in_lds = 1;
      shmem += sb;
    }
  }
  if (p->jit_fn[0]) {  // run-time compiled general kernel with the plan's user expressions
    if (lds_st) return PCG_E_UNSUPPORTED;
    void* argv[1] = {&a};
    const size_t sh = (per_env_t && a.sched_in_lds) ? shmem : integ_shmem;
    return (int)hipModuleLaunchKernel(cov_jit(p->jit_fn[per_env_t ? 1 : 0]), grid_for(io->B, block), 1, 1, block, 1, 1, (unsigned)sh,
                                      (hipStream_t)stream, argv, nullptr);
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
    const int ub = tb(false, p->integrator_id);
    hipLaunchKernelGGL(cov(ufn), dim3(grid_for(io->B, ub)), dim3(ub), sh, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  // Adaptive plans: the work-queue kernel (lanes that finish early pull the next env from an LDS tile).
  // PCG_OPT_VARIANT 1 keeps the classic one-env-per-lane kernel (A/B measurement), PCG_OPT_LDS_STAGES too.
  const bool r4q = is_ros_pair(p->integrator_id);  // either Rosenbrock pair
  const StepFn* qtab = p->integrator_id == PCG_INT_RODAS5 ? k.queue_r5 : r4q ? k.queue_r4 : k.queue;
  const StepFn* q_w1 = p->integrator_id == PCG_INT_RODAS5 ? k.queue_r5w1 : k.queue_r4w1;
  const bool q_forced = p->variant == 5 || std::getenv("PCG_Q_FORCE") != nullptr;  // PCG_OPT_VARIANT 5: any model
  // the launch of a work-queue kernel (geometry, tile, LDS): true = taken, rc_out is the launch's status
  auto queue_launch = [&](StepArgs a, const StepFn* qtab, bool r4q, bool q_forced, int& rc_out) -> bool {
    const int pe = per_env_t ? 1 : 0;
    const size_t sb = (per_env_t && a.sched_in_lds) ? sizeof(double) * (size_t)(c.nsp + c.nd) * c.N : 0;
    rc = queue_geometry(p, k, pe, sb);
    if (rc != PCG_OK) {
      rc_out = rc;
      return true;
    }
    if (p->q_tile[pe] > 0) {
      a.q_tile = p->q_tile[pe];
      if (const char* ev = std::getenv("PCG_Q_TILE")) {  // measurement switch: smaller tile (A/B)
        const int tv = std::atoi(ev);
        if (tv >= QBLOCK && tv <= a.q_tile) a.q_tile = tv;
      }
      if (std::getenv("PCG_Q_NOSORT")) a.q_tile |= 0x10000;  // measurement switch: FIFO order
      if (const char* ev = std::getenv("PCG_Q_REFILL")) a.q_tile |= (std::atoi(ev) & 0x7F) << 20;  // measurement switch

      a.q_w = r4q ? 3.56f : 20.0f;  // (Rodas4: the fit of MEImpl::cost_key_ros) tools/queue_w_sweep.sh: 0 / 10 / 20 / 33 -> me10 0.689 / 0.684 / 0.683 / 0.719 ms, configs[4] shard 0.964 / 0.938 / 0.920 / 0.924 ms
      if (const char* ev = std::getenv("PCG_Q_W")) a.q_w = (float)std::atof(ev);  // measurement switch: key weight
      // Rodas4 with two workgroups per CU: a wave that carries one of the 128 heaviest envs of its tile raises its issue
      // priority (s_setprio) -- it then runs at the speed of a wave that has its SIMD to itself (2.9 instead of 4.7 us per
      // attempt) while its SIMD-mate fills the gaps; the two workgroups of a CU start their heaviest envs on different
      // SIMDs.  configs[4]'s ME segment (349,524 envs: too many for one tile per CU): 484 -> 430 us; no effect on the
      // explicit pair (profiles/r3/queue_prio_sweep.txt).
      a.q_prio = r4q ? 128 : 0;
      if (const char* ev = std::getenv("PCG_Q_PRIO")) a.q_prio = std::atoi(ev);  // measurement switch: issue priority
      // Rodas4, launches of at most ~one full tile per CU (measured: 1024 envs per CU 0.361 -> 0.337 ms; 1366 per CU no
      // difference; 4096 per CU 1.47 -> 1.71 ms): ONE workgroup per CU on the instantiation that keeps the whole loop in
      // registers, every wave alone on its SIMD
      // (the fifth-order pair spills more at two waves per SIMD -- 480 B of scratch per lane against 352 -- and takes the shape up to
      // the 1366 envs per CU of BASELINE configs[4]'s segment, whose lean tile of 1408 slots still fits the CU's LDS with its
      // state: 330 against 337 us per step, HBM traffic 1.06 x the algorithmic bytes against 1.9 x; under the fourth-order pair
      // the same shape was 12 % SLOWER than two workgroups per CU, profiles/r5/mixed_lean_layout.txt, mixed_rodas5.txt)
      int w1_cap = p->integrator_id == PCG_INT_RODAS5 ? 1500 : 1200;  // envs per CU up to which the one-workgroup-per-CU shape is taken
      if (const char* ev = std::getenv("PCG_Q_W1CAP")) w1_cap = std::atoi(ev);  // measurement switch
      bool w1 = r4q && q_w1[pe] && p->q_tile1[pe] >= QBLOCK && io->B <= (int64_t)p->num_cus * w1_cap &&
                io->B > (int64_t)p->num_cus * QBLOCK;
      if (const char* ev = std::getenv("PCG_Q_W1")) w1 = w1 && std::atoi(ev) != 0;  // measurement switch
      if (w1) a.q_tile = (a.q_tile & ~0xFFFF) | p->q_tile1[pe];
      // waves of a workgroup that take part in the cooperative phase of a Rodas4 tile (pcg_step_queue.hpp): the heavy envs of
      // a tile are few (1-2 % at the default threshold) and a wave carries eight at a time -- ONE wave with its groups busy
      // where the workgroup has the CU to itself, two where two workgroups share it (me10_ros4 at B = 2^18: all four waves
      // 326.8 us, two 312.9, one 311.6 at threshold 58; profiles/r5/coop_sweep.txt).  PCG_Q_COOPW: measurement switch (0 = all)
      if (r4q) {
        int cw = w1 ? 1 : 2;
        if (const char* ev = std::getenv("PCG_Q_COOPW")) cw = std::atoi(ev);
        a.q_tile |= (cw & 0xF) << 27;
      }
      // The explicit pair at two waves per SIMD: ONE 512-thread workgroup per CU on a tile of up to 2048 slots instead of two
      // 256-thread workgroups on 1024 each.  The lanes and the envs per lane are the same, the pool is twice as deep, and the
      // two waves of a SIMD drain the same queue: with two workgroups a wave whose SIMD-mate's tile ran dry early finished
      // alone (per-wave stamps, tools/queue_probe.py: the 10-state cascade's waves ended between 424 and 737 us of a 737 us
      // launch).  Taken when every workgroup still gets >= 1.75 envs per lane.
      bool wide = !r4q && !a.fixup && k.queue_w[pe] && io->B >= (int64_t)p->num_cus * (7 * 2 * QBLOCK / 4);
      // (The Rosenbrock pair in this shape -- one 512-thread workgroup per CU, both waves of a SIMD on one tile of 1024 -- was
      // built and measured in round 5: a wave that shares its SIMD takes 6.8 us per attempt, i.e. 3.4 us per wave-attempt
      // against 3.5 alone; 320-323 us per launch against 305-326: declined, profiles/r5/r4wide_sweep.txt.)
      if (const char* ev = std::getenv("PCG_Q_WIDE")) wide = wide && std::atoi(ev) != 0;  // measurement switch
      const int qb = wide ? 2 * QBLOCK : QBLOCK;
      if (wide) {
        int Tw = QSORT;
        while (Tw >= qb && k.queue_lds(Tw) + sb > (size_t)(160 * 1024 - 2048)) Tw -= 64;
        a.q_tile = (a.q_tile & ~0xFFFF) | Tw;
      }
      const int q_bpc = (w1 || wide) ? 1 : p->q_bpc[pe];
      int64_t nwg = (int64_t)p->num_cus * q_bpc;
      const int64_t cap = (io->B + qb - 1) / qb;  // no workgroup with less than one env per lane
      if (nwg > cap) nwg = cap;
      // The queue only pays when its tiles are well filled: with fewer than ~1.75 envs per lane in a sub-tile the
      // re-balancing gain (measured 1.11x at 2.0 on BASELINE configs[2]) no longer covers the bookkeeping (0.99x at
      // 1.33: the ME segment of configs[4]) -- such launches stay on the classic kernel.
      const int64_t per = (io->B + nwg - 1) / nwg;
      int Tq = a.q_tile & 0xFFFF;
      const int64_t nsub = (per + Tq - 1) / Tq;
      const int64_t sub = (per + nsub - 1) / nsub;
      const bool filled = sub >= (7 * qb) / 4 || q_forced;
      if (filled) {
      // LDS for the sub-tile this launch actually walks, not for the largest one the plan could (the kernel derives the
      // same number of sub-tiles from the smaller stride); the tile's state goes to LDS too when that still leaves room
      // for the other workgroups of the CU
      const int Tfit = (int)((sub + 63) / 64 * 64) < qb ? qb : (int)((sub + 63) / 64 * 64);
      if (Tfit < Tq && !std::getenv("PCG_Q_TILE")) {
        Tq = Tfit;
        a.q_tile = (a.q_tile & ~0xFFFF) | Tq;
      }
      // A workgroup that has its CU to itself and whose tile's state does not fit in LDS walks two half tiles that do, as
      // long as a lane still gets two envs (the 20-state cascade at B = 2^18: 1024 envs per CU, 2 x 512 with 80 KB of state
      // each).  Scattered 8-byte accesses of a state that stays in the batch reach HBM as 32-byte sectors -- a tile's rows
      // do not survive in L2 between a lane's pick-up and its neighbours' -- 592 MB per launch against 114 MB of algorithm;
      // from LDS the launch moves 137 MB (1.21 x) and takes 1.7 % longer (two envs per lane instead of four for the
      // longest-first order; profiles/r4/queue_probe/).  PCG_Q_NOXLDS=1 keeps the full tile.
      // (a full tile whose state fits in the LEAN layout -- below -- is preferred to two half tiles)
      const bool lean_fits = !a.fixup && k.queue_lds_x_lean && !std::getenv("PCG_Q_NOXLDS") && !std::getenv("PCG_Q_NOLEAN") &&
                             k.queue_lds_x_lean(Tq, c.na + c.nd) + sb <= (size_t)(160 * 1024 - 2048) / q_bpc;
      if (!lean_fits && q_bpc == 1 && !wide && k.queue_lds_x(Tq) + sb > (size_t)(160 * 1024 - 2048) && !std::getenv("PCG_Q_NOXLDS") &&
          !std::getenv("PCG_Q_TILE")) {
        const int64_t sub2 = (per + 2 * nsub - 1) / (2 * nsub);
        const int T2 = (int)((sub2 + 63) / 64 * 64) < qb ? qb : (int)((sub2 + 63) / 64 * 64);
        if (sub2 >= 2 * qb && k.queue_lds_x(T2) + sb <= (size_t)(160 * 1024 - 2048)) {
          Tq = T2;
          a.q_tile = (a.q_tile & ~0xFFFF) | Tq;
        }
      }
      size_t qsh = k.queue_lds(Tq) + sb;
      const bool force_lean = std::getenv("PCG_Q_FORCE_LEAN") != nullptr;  // test switch: the lean layout wherever it fits
      if (!force_lean && k.queue_lds_x(Tq) + sb <= (size_t)(160 * 1024 - 2048) / q_bpc && !std::getenv("PCG_Q_NOXLDS")) {
        a.q_tile |= 0x20000;
        qsh = k.queue_lds_x(Tq) + sb;
      } else if (lean_fits && k.queue_lds_x_lean(Tq, c.na + c.nd) + sb <= (size_t)(160 * 1024 - 2048) / q_bpc) {  // (Tq may have changed)
        // the LEAN tile layout (pcg_step_queue.hpp, QTile; round 5) where it is what lets the state in: no first-step and
        // step-count arrays, only the configured disturbance values of the held input.  configs[4]'s extraction segment
        // (349,524 envs = 683 per workgroup, two workgroups per CU: 98.8 KB each in the full layout, 76.2 in this one) moved
        // 5.1 x its algorithmic bytes with its state in the batch (profiles/r5/pmc.json)
        a.q_tile |= 0x20000 | 0x40000;
        qsh = k.queue_lds_x_lean(Tq, c.na + c.nd) + sb;
      }
      if (a.fixup) {  // fix-up launch: a tile is a compact list of up to Tq MARKED envs (+ 4 bytes per slot: which env), the
        // state stays in the batch
        Tq = p->q_tile[pe];
        while (Tq > qb && k.queue_lds(Tq) + 4 * (size_t)Tq + 8 + sb > (size_t)(160 * 1024 - 2048) / q_bpc) Tq -= 64;
        if (Tq < qb) {  // (the scan of the fix-up kernel parks QB envs per round: a tile smaller than that never advances)
          rc_out = PCG_E_UNSUPPORTED;
          return true;
        }
        a.q_tile = (a.q_tile & ~(0xFFFF | 0x20000)) | Tq;
        qsh = k.queue_lds(Tq) + 4 * (size_t)Tq + 8 + sb;
      }
      hipLaunchKernelGGL(cov(w1 ? q_w1[pe] : wide ? k.queue_w[pe] : qtab[pe]), dim3((unsigned)nwg), dim3(qb), qsh, (hipStream_t)stream, a);
      rc_out = (int)hipGetLastError();
      return true;
      }
    }
      return false;
  };
  if ((p->integrator_id == PCG_INT_DOPRI5 || r4q) && !lds_st && (p->variant == 0 || p->variant == 5) && qtab[per_env_t ? 1 : 0] &&
      (k.queue_default || q_forced)) {
    int qrc = PCG_OK;
    if (queue_launch(a, qtab, r4q, q_forced, qrc)) return qrc;
  }
  // Guarded plans (PCG_INT_RK4G / PCG_INT_T5G) in TWO launches: the general kernel takes the guarded fixed step of every
  // env and only MARKS the ones it does not trust (done[e] = 2, nothing else of their step stored); the work-queue kernel
  // of the adaptive pair then integrates exactly the marked envs -- longest first, lanes pulling the next one -- and
  // finishes their step.  In one launch the fallback ran inside the wave that met it: with a third of a batch igniting
  // every wave waited for its slowest lane (607 us per 2^20-env step on the full x0 box of the cstr against 362 us for the
  // adaptive pair through the queue alone).  Same arithmetic per env either way (tests: the oracle's t5g / rk4g twins).
  // Costs the calm closed loop one nearly empty launch.  Not with a_delta (env_pre accumulates into a_save: not idempotent).
  bool fixup = (p->integrator_id == PCG_INT_RK4G || p->integrator_id == PCG_INT_T5G) && !lds_st && p->variant == 0 &&
               k.queue_fix[per_env_t ? 1 : 0] && !(c.flags & PCG_F_A_DELTA) && io->B >= (int64_t)p->num_cus * QBLOCK &&
               !std::getenv("PCG_NO_FIXUP");
  if (fixup) {
    const size_t sbq = (per_env_t && a.sched_in_lds) ? sizeof(double) * (size_t)(c.nsp + c.nd) * c.N : 0;
    rc = queue_geometry(p, k, per_env_t ? 1 : 0, sbq);
    if (rc != PCG_OK) return rc;
    fixup = p->q_tile[per_env_t ? 1 : 0] > 0;
  }
  a.fixup = fixup ? 1 : 0;
  // lean variant when no noise / Gaussian disturbance / constraint work is configured
  // (the lean kernels also compile out a_delta, the terminal "batch" reward and per-env disturbances)
  const bool extras = (c.flags & (PCG_F_NOISE | PCG_F_GAUSS_DIST | PCG_F_A_DELTA | PCG_F_REWARD_BATCH | PCG_F_REWARD_TRACK)) ||
                      c.ncon > 0 || io->d != nullptr;
  // streaming (persistent, prefetching, 16 B/lane) kernel for the lean lock-stepped path
  // Adaptive stepping is left to the one-wave-per-workgroup classic kernel unless a streaming variant is forced:
  // lanes take different numbers of steps, and a persistent grid fixes each wave's share of the batch up front,
  // whereas the dispatcher hands single-wave workgroups to whichever SIMD slot frees first.
  const int ls = lean_scheme(p->integrator_id);  // fixed-step schemes with a lean pipelined kernel (RK4, CV8)
  const StepFn* const pipe = ls >= 0 ? k.pipe[ls] : nullptr;
  const bool stream_ok = ls >= 0 || p->variant == 2 || p->variant == 3;
  const bool lean_ar_ok = !auto_reset || ((p->variant == 4 || p->variant == 0 || p->variant == 5) && pipe && pipe[0]);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  auto al2 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 1u) == 0; };
  const bool pipe_ok = (p->variant == 4 || p->variant == 0 || p->variant == 5) && pipe && pipe[0];
  // (the lean kernels index envs and row offsets in 32 bits: B < 2^28; larger batches take the classic kernel)
  if (!per_env_t && !extras && !lds_st && !io->viol && (!io->status || pipe_ok) && p->variant != 1 && stream_ok && lean_ar_ok &&
      io->B < ((int64_t)1 << 28) && (k.stream[p->integrator_id][0] || pipe_ok)) {
    const bool epl2_ok = (k.stream[p->integrator_id][1] || (pipe_ok && pipe[1])) && (io->B % 2 == 0) && al16(io->x) &&
                         al16(io->a) && al16(io->obs) && al16(io->rew) && al2(io->done);
    int epl = (p->variant == 2) ? 1 : (epl2_ok ? 2 : 1);
    if (p->variant == 3 && !epl2_ok) return PCG_E_UNSUPPORTED;
    if (const char* ev = std::getenv("PCG_LEAN_EPL"))  // measurement switch: one env per lane
      if (std::atoi(ev) == 1) epl = 1;
    StepFn sfn = k.stream[p->integrator_id][epl - 1];
    // auto (0): the software-pipelined kernel where it exists (measured best on the cstr workload:
    // 14.9 us vs 15.0 two-sub-tile streaming vs 16.9 plain streaming vs 21 classic, profiles/r1)
    const bool piped = pipe_ok && pipe[epl - 1];
    if (piped) sfn = auto_reset ? k.pipe_ar[ls][epl - 1] : pipe[epl - 1];
    if (!sfn) return PCG_E_UNSUPPORTED;  // (a forced streaming variant of a scheme that only has the pipelined kernel)
    int& occ = piped ? p->pipe_occ[auto_reset ? 1 : 0][epl - 1] : p->stream_occ[epl - 1];
    if (occ == 0) {
      const int q = resident_blocks(sfn);
      if (q < 0) return -q;
      occ = q;
    }
    const int64_t tile_envs = (int64_t)BLOCK * epl;
    const int64_t ntile = (io->B + tile_envs - 1) / tile_envs;
    int bpc = occ;
    // HBM-bound lean kernels: five resident workgroups per CU.  More waves per SIMD only lengthen every wave's integration
    // phase (they share the vector unit round-robin), so the grid's stores leave later and in a shorter burst -- measured on
    // the cstr headline, interleaved runs of one box (profiles/r4/headline_bisect.txt): 4 / 5 / 6 / 8 per CU = 13.96 /
    // 12.60 / 13.58 / 15.1 us.  PCG_OPT_STREAM_BLOCKS_PER_CU overrides.
    if (piped && bpc > 5) bpc = 5;
    if (p->stream_bpc > 0 && p->stream_bpc <= occ) bpc = p->stream_bpc;
    int64_t grid = (int64_t)p->num_cus * bpc;
    if (grid > ntile) grid = ntile;
    a.nt_stores = p->nt_stores;
    if (piped) {  // issue priority by residency slot (step_kernel_pipe); PCG_LEAN_PRIO: measurement override (hex, 2 bits per slot)
      static const int prio_env = [] { const char* ev = std::getenv("PCG_LEAN_PRIO"); return ev ? (int)std::strtol(ev, nullptr, 16) : -1; }();
      a.q_prio = prio_env >= 0 ? prio_env : 0;
      a.q_tile = p->num_cus;
    }
    hipLaunchKernelGGL(cov(sfn), dim3((unsigned)grid), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  // Feature-masked pipelined kernel (pcg_step_feat.hpp): RK4 plans of the small models with anything beyond the
  // lean step switched on.  Needs two envs per lane (even B, 16-byte rows); the smallest instantiation whose mask
  // covers what this launch uses is taken.  PCG_OPT_VARIANT 1 forces the classic one-env-per-lane kernel (A/B).
  if (p->integrator_id == PCG_INT_RK4 && k.nfeat > 0 && !lds_st && (p->variant == 0 || p->variant == 4 || p->variant == 5)) {
    unsigned need = 0;
    if (c.ncon > 0) need |= FT_CONS;
    if (c.flags & PCG_F_A_DELTA) need |= FT_ADELTA;
    if (c.flags & PCG_F_REWARD_TRACK) need |= FT_TRACK;
    if (c.flags & PCG_F_REWARD_BATCH) need |= FT_BATCH;
    if (auto_reset) need |= FT_AR;
    // observation noise, per-env step counters, per-env / Gaussian disturbances: the classic kernel is the faster one
    // (measured), and a
    // lock-stepped same-launch reset needs every env to end together
    bool ok = !per_env_t && !io->d && !(c.flags & PCG_F_NOISE) && !((c.flags & PCG_F_GAUSS_DIST) && c.nd > 0) &&
              !(auto_reset && (c.flags & PCG_F_DONE_ON_CONS) && c.ncon > 0) && (io->B % 2 == 0) && al16(io->x) &&
              al16(io->a) && al16(io->obs) && al16(io->rew) && al2(io->done) && al2(io->viol) && al2(io->status) &&
              al16(io->a_save) && al16(io->u_prev) && al16(io->g) && al16(io->g_pre);
    int best = -1;
    for (int i = 0; ok && i < k.nfeat; ++i)
      if ((k.feat[i].mask & need) == need &&
          (best < 0 || __builtin_popcount(k.feat[i].mask) < __builtin_popcount(k.feat[best].mask)))
        best = i;
    if (best >= 0) {
      if (p->feat_occ[best] == 0) {
        const int q = resident_blocks(k.feat[best].fn);
        if (q < 0) return -q;
        p->feat_occ[best] = q;
      }
      const int64_t tile_envs = (int64_t)BLOCK * 2;
      const int64_t ntile = (io->B + tile_envs - 1) / tile_envs;
      int bpc = p->feat_occ[best];
      if (p->stream_bpc > 0 && p->stream_bpc < bpc) bpc = p->stream_bpc;
      int64_t grid = (int64_t)p->num_cus * bpc;
      if (grid > ntile) grid = ntile;
      a.nt_stores = p->nt_stores;
      hipLaunchKernelGGL(cov(k.feat[best].fn), dim3((unsigned)grid), dim3(BLOCK), 0, (hipStream_t)stream, a);
      return (int)hipGetLastError();
    }
  }
  StepFn fn = k.step[p->integrator_id][per_env_t ? 1 : 0][lds_st ? 1 : 0][extras ? 1 : 0];
  if (!fn) return PCG_E_UNSUPPORTED;
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(cov(fn), dim3(grid_for(io->B, block)), dim3(block), shmem, (hipStream_t)stream, a);
  rc = (int)hipGetLastError();
  if (rc != PCG_OK || !fixup) return rc;
  int qrc = PCG_OK;
  if (!queue_launch(a, k.queue_fix, false, true, qrc)) return PCG_E_UNSUPPORTED;  // (not reachable: the geometry was checked above)
  return qrc;
}

int pcg_step(pcg_plan* p, const pcg_buffers* io, int32_t t, uint64_t seed, void* stream) {
  return step_impl(p, io, t, seed, stream, false, 0);
}

int pcg_step_autoreset(pcg_plan* p, const pcg_buffers* io, int32_t t, uint64_t seed, uint64_t reset_seed, void* stream) {
  if (plan_ok(p) && io && p->hc.nunc > 0 && !io->p_unc) return PCG_E_NULL;
  return step_impl(p, io, t, seed, stream, true, reset_seed);
}

int pcg_rollout_strided(pcg_plan* p, const pcg_buffers* io, int32_t t0, int32_t T, const double* a_seq,
                        int64_t a_step_stride, int64_t a_comp_stride, double* obs_seq, int64_t obs_step_stride,
                        int64_t obs_comp_stride, double* rew_seq, int64_t rew_step_stride, uint64_t seed,
                        void* stream) {
  StepArgs a;
  int rc = fill_args(p, io, &a);
  if (rc != PCG_OK) return rc;
  if (io->t) return PCG_E_UNSUPPORTED;  // lock-stepped only
  if (T < 1 || t0 < 0 || (int64_t)t0 + (int64_t)T > 0x7fffffffLL) return PCG_E_VALUE;  // (t0 indexes the per-step tables)
  if (io->B == 0) return PCG_OK;
  if (!io->x || !a_seq || !io->obs || !io->rew || !io->done) return PCG_E_NULL;
  const DevConst& c = p->hc;
  if ((c.flags & PCG_F_A_DELTA) && !io->a_save) return PCG_E_NULL;
  if ((c.flags & PCG_F_REWARD_TRACK) && !io->u_prev) return PCG_E_NULL;
  if (p->jit_fn[0] && !p->jit_roll) return PCG_E_UNSUPPORTED;
  if (c.nunc > 0) {  // per-env parameters (sampled by the reset before the episode): the general rollout kernel's UNC form
    if (!io->p_unc) return PCG_E_NULL;
    const Kernels& ku = kernels(p->kid);
    const StepFn ufn = p->jit_fn[0] ? nullptr : ku.rollout_unc[p->integrator_id];
    if (!ufn) return PCG_E_UNSUPPORTED;  // RK4 and the explicit pair only, as for stepping
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
    const int ub = tb(false, p->integrator_id);
    hipLaunchKernelGGL(cov(ufn), dim3(grid_for(io->B, ub)), dim3(ub), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
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
  if (p->jit_fn[0]) {  // run-time compiled rollout kernel with the plan's user expressions (general step, one env per lane)
    void* argv[1] = {&a};
    const int jb = tb(false, p->integrator_id);
    return (int)hipModuleLaunchKernel(cov_jit(p->jit_roll), grid_for(io->B, jb), 1, 1, jb, 1, 1, 0, (hipStream_t)stream, argv, nullptr);
  }
  const Kernels& k = kernels(p->kid);
  const bool lds_st = p->lds_stages && p->integrator_id == PCG_INT_DOPRI5 && k.has_lds_stages;
  const bool extras = (c.flags & (PCG_F_NOISE | PCG_F_GAUSS_DIST | PCG_F_A_DELTA | PCG_F_REWARD_BATCH | PCG_F_REWARD_TRACK)) ||
                      c.ncon > 0 || io->d != nullptr;
  if (!extras && p->integrator_id == PCG_INT_RK4 && !io->viol && p->variant != 1 && k.roll_lean[0]) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    const bool ev = ((a.a_ss | a.a_cs | a.o_ss | a.o_cs | a.r_ss) & 1) == 0;  // 16-byte rows stay 16-byte aligned
    const bool e2 = ev && k.roll_lean[1] && (io->B % 2 == 0) && al16(io->x) && al16(a_seq) && al16(io->obs) &&
                    al16(io->rew) && (!obs_seq || al16(obs_seq)) && (!rew_seq || al16(rew_seq)) &&
                    (reinterpret_cast<uintptr_t>(io->done) & 1u) == 0;
    const int epl = e2 ? 2 : 1;
    hipLaunchKernelGGL(cov(k.roll_lean[epl - 1]), dim3(grid_for(io->B, BLOCK * epl)), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  const int block = tb(lds_st, p->integrator_id);
  const size_t shmem = lds_st ? sizeof(double) * 6 * (size_t)k.nx * BLOCK_LDS : 0;
  StepFn fn = k.rollout[p->integrator_id][lds_st ? 1 : 0];
  if (!fn) return PCG_E_UNSUPPORTED;  // the Rosenbrock integrator steps through pcg_step only
  // The guarded default plan of a model with a guard (PCG_INT_T5G), batches of at least one wave per SIMD: the barrier-free
  // rollout in two passes (pcg_rollout_flat.hpp).  The first pass is this plan's ordinary fused rollout kernel, told to hand
  // an env over at the first step its guard does not trust; the second carries each handed-over env to the end of the rollout
  // on a lane of its own.  Same bits as T pcg_step launches.  Not with a_delta (env_pre accumulates), not while the stream is
  // being captured into a graph before the work space exists (hipMalloc); PCG_NO_FLAT=1 keeps the single-kernel rollout (A/B).
  if (p->integrator_id == PCG_INT_T5G && k.roll_hot && !lds_st && p->variant == 0 && !(c.flags & PCG_F_A_DELTA) && T >= 2 &&
      io->B >= (int64_t)p->num_cus * 4 * 64 && io->B < ((int64_t)1 << 31) && !std::getenv("PCG_NO_FLAT")) {
    if (p->flat_cap < io->B) {
      if (p->flat_ws) HIP_TRY(hipFree(p->flat_ws));
      p->flat_ws = nullptr;
      p->flat_cap = 0;
      HIP_TRY(hipMalloc((void**)&p->flat_ws, sizeof(int32_t) * (4 + 2 * (size_t)io->B)));
      p->flat_cap = io->B;
    }
    HIP_TRY(hipMemsetAsync(p->flat_ws, 0, sizeof(int32_t) * 4, (hipStream_t)stream));
    a.flat_q = p->flat_ws;
    a.flat_hot = p->flat_ws + 4;
    a.flat_tstar = p->flat_ws + 4 + p->flat_cap;
    a.fixup = 1;
    hipLaunchKernelGGL(cov(fn), dim3(grid_for(io->B, block)), dim3(block), shmem, (hipStream_t)stream, a);
    rc = (int)hipGetLastError();
    if (rc != PCG_OK) return rc;
    int wps = 3;  // persistent waves per SIMD of the second pass (measured: profiles/r6/flat_rollout.txt)
    if (const char* ev = std::getenv("PCG_FLAT_WPS")) wps = std::max(1, std::min(8, std::atoi(ev)));  // measurement switch
    a.q_tile = 2;  // ... and the cadence of its step boundaries (rollout_kernel_hot: `every`)
    if (const char* ev = std::getenv("PCG_FLAT_EVERY")) a.q_tile = std::max(1, std::min(64, std::atoi(ev)));  // measurement switch
    hipLaunchKernelGGL(cov(k.roll_hot), dim3((unsigned)(p->num_cus * wps)), dim3(FLAT_BLOCK), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(cov(fn), dim3(grid_for(io->B, block)), dim3(block), shmem, (hipStream_t)stream, a);
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
  hipLaunchKernelGGL(cov(reset_kernel), dim3(grid_for(io->B)), dim3(BLOCK), 0, (hipStream_t)stream, a);
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
  if (io->t || p->jit_fn[0]) return PCG_E_UNSUPPORTED;
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
    (void)hipStreamDestroy(cs);
    return (int)e;
  }
  if (with_reset) rc = pcg_reset(p, &b, nullptr, seed, cs);
  for (int j = 0; j < T && rc == PCG_OK; ++j) {
    b.a = a_steps[j];
    b.d = d_steps ? d_steps[j] : nullptr;
    rc = pcg_step(p, &b, t0 + j, seed, cs);
  }
  e = hipStreamEndCapture(cs, &g);
  (void)hipStreamDestroy(cs);
  if (rc != PCG_OK) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    return (int)e;
  }
  pcg_graph* q = new (std::nothrow) pcg_graph();
  if (!q) {
    (void)hipGraphExecDestroy(ex);
    (void)hipGraphDestroy(g);
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
  (void)hipGraphExecDestroy(q->exec);
  (void)hipGraphDestroy(q->graph);
  q->magic = 0;
  delete q;
  return PCG_OK;
}

int pcg_rhs(pcg_plan* p, int64_t B, const double* x, const double* u, double* dx, void* stream) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!x || !u || !dx) return PCG_E_NULL;
  if (B <= 0) return B == 0 ? PCG_OK : PCG_E_DIM;
  if (p->jit_rhs) {  // PCG_MODEL_USER: the run-time compiled hook
    const PCG_CONSTANT DevConst* dc = (CDevConst*)p->dC;
    int nu = p->cfg_nu;
    void* argv[6] = {&dc, &B, &nu, &x, &u, &dx};
    return (int)hipModuleLaunchKernel(cov_jit(p->jit_rhs), grid_for(B), 1, 1, BLOCK, 1, 1, 0, (hipStream_t)stream, argv, nullptr);
  }
  if (p->model_id == PCG_MODEL_USER) return PCG_E_PLAN;
  const Kernels& k = kernels(p->kid);
  hipLaunchKernelGGL(cov(k.rhs), dim3(grid_for(B)), dim3(BLOCK), 0, (hipStream_t)stream, (CDevConst*)p->dC, B, p->cfg_nu, x, u, dx);
  return (int)hipGetLastError();
}

int pcg_integrate(pcg_plan* p, int64_t B, double* x, const double* u, int32_t* nsteps, void* stream) {
  if (!plan_ok(p)) return PCG_E_PLAN;
  if (!x || !u) return PCG_E_NULL;
  if (B <= 0) return B == 0 ? PCG_OK : PCG_E_DIM;
  if (p->jit_integ) {  // PCG_MODEL_USER: the run-time compiled hook
    const int ub = tb(false, p->integrator_id, p->nx);
    const size_t ush = sizeof(double) * integ_lds_doubles(p->nx, p->integrator_id, false);
    const PCG_CONSTANT DevConst* dc = (CDevConst*)p->dC;
    int nu = p->cfg_nu;
    void* argv[6] = {&dc, &B, &nu, &x, &u, &nsteps};
    return (int)hipModuleLaunchKernel(cov_jit(p->jit_integ), grid_for(B, ub), 1, 1, ub, 1, 1, (unsigned)ush, (hipStream_t)stream, argv,
                                      nullptr);
  }
  if (p->model_id == PCG_MODEL_USER) return PCG_E_PLAN;
  const Kernels& k = kernels(p->kid);
  const bool lds_st = p->lds_stages && p->integrator_id == PCG_INT_DOPRI5 && k.has_lds_stages;
  const int block = tb(lds_st, p->integrator_id, k.nx, k.ros_structured);
  const size_t shmem = sizeof(double) * integ_lds_doubles(k.nx, p->integrator_id, lds_st, k.ros_structured);
  IntKFn fn = k.integ[p->integrator_id][lds_st ? 1 : 0];
  if (!fn) return PCG_E_UNSUPPORTED;
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(cov(fn), dim3(grid_for(B, block)), dim3(block), shmem, (hipStream_t)stream, (CDevConst*)p->dC, B, p->cfg_nu, x,
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
