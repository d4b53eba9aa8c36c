/* user_model_demo.c -- a user model through the C ABI from plain C: PCG_MODEL_USER with its right-hand side given as C
 * statements (pcg_env_cfg.user_rhs_src, compiled into the plan's kernels by pcg_plan_create with hipRTC).
 *
 * A Monod chemostat with substrate inhibition: states X, S; input D (dilution rate); parameters mumax, Ks, Ki, Y, Sf.
 * B environments, adaptive integrator, set-point tracking on X, a scripted action sequence.  Prints one line per checked
 * quantity; tests/test_c_host.py compares them with pcgym_amd.VecEnv on the same configuration.
 *
 * build:  gcc -std=c11 -O2 -I/opt/rocm/include -I../../include user_model_demo.c -L../../pc-gym_amd -lpcgym_hip \
 *             -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/../../pc-gym_amd -Wl,-rpath,/opt/rocm/lib -o user_model_demo
 * run:    ./user_model_demo <B> <directory of the library's kernel headers = pc-gym_amd/csrc>
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcgym_hip.h"

#define CK(call)                                                              \
  do {                                                                        \
    int rc_ = (int)(call);                                                    \
    if (rc_ != 0) {                                                           \
      fprintf(stderr, "%s failed: %d (%s)\n%s\n", #call, rc_, pcg_strerror(rc_), pcg_last_jit_log()); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <B> <kernel header directory>\n", argv[0]);
    return 2;
  }
  const int64_t B = atoll(argv[1]);
  enum { N = 30, NX = 2, NA = 1, NOBS = 3 };
  /* the model: what a Python user writes as custom_model = {"states", "inputs", "parameters", "aux", "rhs"} */
  const double params[5] = {0.53, 0.12, 22.0, 0.4, 4.0}; /* mumax Ks Ki Y Sf */
  const char* rhs =
      "  const double mu = p[0]*x[1]/(p[1] + x[1] + x[1]*x[1]/p[2]);\n"
      "  dx[0] = (mu - u[0])*x[0];\n"
      "  dx[1] = u[0]*(p[4] - x[1]) - mu*x[0]/p[3];";
  double sp[N];
  for (int i = 0; i < N; ++i) sp[i] = i < N / 2 ? 1.4 : 1.0;
  const double x0[3] = {1.2, 0.6, 1.4}, a_low[1] = {0.0}, a_high[1] = {0.45};
  const double o_low[3] = {0.0, 0.0, 0.0}, o_high[3] = {3.0, 6.0, 3.0}, r_scale[1] = {10.0};
  const int32_t sp_index[1] = {0};

  pcg_env_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.model_id = PCG_MODEL_USER;
  cfg.integrator_id = PCG_INT_DOPRI5;
  cfg.nx = NX; cfg.na = NA; cfg.nsp = 1; cfg.nsp_obs = 1; cfg.N = N;
  cfg.substeps = 8; cfg.max_steps = 100000;
  cfg.flags = PCG_F_NORMALISE_A | PCG_F_NORMALISE_O | PCG_F_MAXIMISE | PCG_F_REF_COMPAT;
  cfg.n_params = 5;
  cfg.dt = 15.0 / N; cfg.rtol = 1e-8; cfg.atol = 1e-8;
  cfg.params = params; cfg.x0 = x0; cfg.a_low = a_low; cfg.a_high = a_high;
  cfg.o_low = o_low; cfg.o_high = o_high; cfg.sp_index = sp_index; cfg.sp = sp; cfg.r_scale = r_scale;
  cfg.user_rhs_src = rhs;
  cfg.jit_include_dir = argv[2];
  CK(pcg_cfg_validate(&cfg));
  pcg_plan* plan = NULL;
  CK(pcg_plan_create(&plan, &cfg));

  pcg_buffers io;
  memset(&io, 0, sizeof io);
  io.B = B;
  double* d_a = NULL;
  CK(hipMalloc((void**)&io.x, sizeof(double) * NX * B));
  CK(hipMalloc((void**)&d_a, sizeof(double) * NA * B));
  CK(hipMalloc((void**)&io.obs, sizeof(double) * NOBS * B));
  CK(hipMalloc((void**)&io.rew, sizeof(double) * B));
  CK(hipMalloc((void**)&io.done, B));
  CK(hipMalloc((void**)&io.nsteps, sizeof(int32_t) * 2 * B));
  CK(hipMalloc((void**)&io.status, B));
  CK(hipMemset(io.status, 0, B));
  io.a = d_a;

  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  double* h_a = (double*)malloc(sizeof(double) * B);
  double* h = (double*)malloc(sizeof(double) * NOBS * B);
  uint8_t* h_b = (uint8_t*)malloc(B);
  double ret = 0.0;
  CK(pcg_reset(plan, &io, NULL, /*seed*/ 1, stream));
  for (int t = 0; t < N - 1; ++t) {
    for (int64_t e = 0; e < B; ++e) h_a[e] = -1.0 + 2.0 * (double)((e * 5 + t * 11) % 97) / 96.0; /* scripted policy */
    CK(hipMemcpyAsync(d_a, h_a, sizeof(double) * B, hipMemcpyHostToDevice, stream));
    CK(pcg_step(plan, &io, t, /*seed*/ 1, stream));
    CK(hipMemcpyAsync(h, io.rew, sizeof(double) * B, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    for (int64_t e = 0; e < B; ++e) ret += h[e];
  }
  CK(hipMemcpy(h, io.obs, sizeof(double) * NOBS * B, hipMemcpyDeviceToHost));
  double s_obs[NOBS] = {0, 0, 0};
  for (int i = 0; i < NOBS; ++i)
    for (int64_t e = 0; e < B; ++e) s_obs[i] += h[(size_t)i * B + e];
  CK(hipMemcpy(h_b, io.status, B, hipMemcpyDeviceToHost));
  int64_t n_bad = 0;
  for (int64_t e = 0; e < B; ++e) n_bad += h_b[e] != PCG_ST_OK;
  printf("return_sum %.17g\n", ret);
  printf("obs_sum %.17g %.17g %.17g\n", s_obs[0], s_obs[1], s_obs[2]);
  printf("obs_env0 %.17g %.17g %.17g\n", h[0], h[B], h[2 * B]);
  printf("n_failed %lld\n", (long long)n_bad);
  /* a source that does not compile comes back as PCG_E_JIT with the compiler's text, never as a crash */
  cfg.user_rhs_src = "  dx[0] = nonsense(x[0]);\n  dx[1] = 0.0;";
  pcg_plan* bad = NULL;
  const int rc = pcg_plan_create(&bad, &cfg);
  printf("bad_source_status %d\n", rc);
  printf("bad_source_log_mentions_nonsense %d\n", strstr(pcg_last_jit_log(), "nonsense") != NULL);

  CK(pcg_plan_destroy(plan));
  hipFree(io.x); hipFree(d_a); hipFree(io.obs); hipFree(io.rew); hipFree(io.done); hipFree(io.nsteps); hipFree(io.status);
  free(h_a); free(h); free(h_b);
  return 0;
}
