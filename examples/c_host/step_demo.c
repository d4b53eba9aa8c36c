/* step_demo.c -- the C ABI of include/pcgym_hip.h driven from plain C: no Python, no torch.
 *
 * pc-gym's quick-start CSTR configuration (reference README.md:16-55: N = 60, tsim = 26, SP on Ca in thirds,
 * normalised actions / observations, r_scale Ca = 1e3), B environments, a fixed action sequence, N-1 steps.
 * Prints one line per checked quantity; tests/test_c_host.py compares them with pcgym_amd.VecEnv on the same inputs.
 *
 * build:  gcc -std=c11 -O2 -I/opt/rocm/include -I../../include step_demo.c -L../../pc-gym_amd -lpcgym_hip \
 *             -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/../../pc-gym_amd -Wl,-rpath,/opt/rocm/lib -o step_demo
 * (plain gcc: the HIP runtime is only needed for hipMalloc / hipMemcpy / streams of the caller-owned buffers).
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
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, pcg_strerror(rc_)); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  enum { N = 60, NX = 2, NA = 1, NOBS = 3 };
  double params[16];
  int32_t nx, nu, ndm, npar;
  CK(pcg_model_info(PCG_MODEL_CSTR, &nx, &nu, &ndm, &npar));
  CK(pcg_model_default_params(PCG_MODEL_CSTR, params, npar));

  double sp[N];
  for (int i = 0; i < N; ++i) sp[i] = i < N / 3 ? 0.85 : (i < 2 * (N / 3) ? 0.9 : 0.87);
  const double x0[3] = {0.8, 330.0, 0.8}, a_low[1] = {295.0}, a_high[1] = {302.0};
  const double o_low[3] = {0.7, 300.0, 0.8}, o_high[3] = {1.0, 350.0, 0.9}, r_scale[1] = {1e3};
  const int32_t sp_index[1] = {0};

  pcg_env_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.model_id = PCG_MODEL_CSTR;
  cfg.integrator_id = PCG_INT_RK4;
  cfg.nx = NX; cfg.na = NA; cfg.nsp = 1; cfg.nsp_obs = 1; cfg.N = N;
  cfg.substeps = 4;  /* dt = 26/60: the reference's accuracy class with 4 RK4 sub-steps */
  cfg.max_steps = 10000;
  cfg.flags = PCG_F_NORMALISE_A | PCG_F_NORMALISE_O | PCG_F_MAXIMISE | PCG_F_REF_COMPAT;
  cfg.n_params = npar;
  cfg.dt = 26.0 / N; cfg.rtol = 1e-8; cfg.atol = 1e-8;
  cfg.params = params; cfg.x0 = x0; cfg.a_low = a_low; cfg.a_high = a_high;
  cfg.o_low = o_low; cfg.o_high = o_high; cfg.sp_index = sp_index; cfg.sp = sp; cfg.r_scale = r_scale;
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
  io.a = d_a;
  printf("bytes_per_env_step %lld\n", (long long)pcg_plan_bytes_per_env_step(plan, &io));

  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  double* h_a = (double*)malloc(sizeof(double) * B);
  double* h = (double*)malloc(sizeof(double) * NOBS * B);
  uint8_t* h_done = (uint8_t*)malloc(B);
  double ret = 0.0;

  CK(pcg_reset(plan, &io, NULL, /*seed*/ 1, stream));
  for (int t = 0; t < N - 1; ++t) {
    for (int64_t e = 0; e < B; ++e) h_a[e] = -1.0 + 2.0 * (double)((e * 7 + t * 13) % 101) / 100.0; /* scripted policy */
    CK(hipMemcpyAsync(d_a, h_a, sizeof(double) * B, hipMemcpyHostToDevice, stream));
    CK(pcg_step(plan, &io, t, /*seed*/ 1, stream));
    CK(hipMemcpyAsync(h, io.rew, sizeof(double) * B, hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    for (int64_t e = 0; e < B; ++e) ret += h[e];
  }
  CK(hipMemcpy(h, io.obs, sizeof(double) * NOBS * B, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_done, io.done, B, hipMemcpyDeviceToHost));
  double s_obs[NOBS] = {0, 0, 0};
  int64_t n_done = 0;
  for (int i = 0; i < NOBS; ++i)
    for (int64_t e = 0; e < B; ++e) s_obs[i] += h[(size_t)i * B + e];
  for (int64_t e = 0; e < B; ++e) n_done += h_done[e];
  printf("return_sum %.17g\n", ret);
  printf("obs_sum %.17g %.17g %.17g\n", s_obs[0], s_obs[1], s_obs[2]);
  printf("obs_env0 %.17g %.17g %.17g\n", h[0], h[B], h[2 * B]);
  printf("n_done %lld\n", (long long)n_done);

  CK(pcg_plan_destroy(plan));
  hipFree(io.x); hipFree(d_a); hipFree(io.obs); hipFree(io.rew); hipFree(io.done);
  free(h_a); free(h); free(h_done);
  return 0;
}
