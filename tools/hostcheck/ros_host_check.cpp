// ros_host_check.cpp -- the Rosenbrock attempt templates of the product headers, compiled for the HOST under sanitizers.
//
// Round 5 found step_kernel<heat_exchanger (24 states), RODAS4 | RODAS5, lock-stepped> returning a garbage x[2]; the fix was
// to write the attempt of models with more than 16 states as loops that stay loops (pcg_integrators.hpp: ros_try_rolled) --
// chosen because the tests pass there.  "Wrong at -O3, right at -O1, moves with the spill strategy" is the signature of a
// compiler problem AND of undefined behaviour in the source (an uninitialised element, an out-of-range index of a private
// array).  This unit decides between the two for the SOURCE: the same templates -- rodas4_try / rodas5_try fully unrolled
// at NX = 24 and 16, ros_try_rolled, RosDense (difference-quotient Jacobian + pivoted LU in "LDS") and the controller loop
// ros_pair -- run on the host
//   * under -fsanitize=address,undefined (gcc): every index of a private array and of the per-lane LDS matrices checked,
//     signed overflow / invalid shifts / float-cast overflow reported;
//   * under clang's -ftrivial-auto-var-init=pattern and =zero: a result that depended on an uninitialised local differs
//     between the two builds (the digests printed at the end must be equal);
//   * unrolled against rolled, bit for bit, on every lane.
// Build and run: tools/hostcheck/run.sh   (g++ / the ROCm clang++; no GPU, no HIP runtime)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// ---- stand-ins for what the device compiler provides -------------------------------------------------------------------
#define __HIPCC_RTC__ 1  // keeps <hip/hip_runtime.h> out (the headers' own switch for run-time compilation)
#define __device__
#define __host__
#define __forceinline__ inline
struct HostIdx { unsigned x, y, z; };
static HostIdx threadIdx{0, 0, 0};  // RosLds addresses lane threadIdx.x of the per-lane matrices
static inline long long __double_as_longlong(double v) { long long r; std::memcpy(&r, &v, 8); return r; }
static inline double __longlong_as_double(long long v) { double r; std::memcpy(&r, &v, 8); return r; }
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
#define __builtin_amdgcn_exp2f(x) std::exp2((float)(x))
#define __builtin_amdgcn_logf(x) std::log2((float)(x))
static inline double host_frexp_mant(double x) { int e; return std::frexp(x, &e); }
static inline int host_frexp_exp(double x) { int e; (void)std::frexp(x, &e); return e; }
#define __builtin_amdgcn_frexp_mant(x) host_frexp_mant(x)
#define __builtin_amdgcn_frexp_exp(x) host_frexp_exp(x)
using std::ldexp;

#define PCG_HOST_CHECK 1
#ifndef PCG_ROS_ROLLED_ABOVE
#define PCG_ROS_ROLLED_ABOVE 9999  // ros_pair_try takes the UNROLLED attempt whatever NX (the form that broke on the device),
#endif                             // ros_solve its select-chain pivot exchanges; -DPCG_ROS_ROLLED_ABOVE=12: the product's forms
#include "../../pc-gym_amd/csrc/pcg_integrators.hpp"

using namespace pcg;

template <class M>
struct HostRhs {  // pcg_kernels.hpp: RhsFn, with the constants in ordinary memory
  const typename M::KP& kp;
  const typename M::Hold& hold;
  void operator()(const double (&x)[M::NX], double (&dx)[M::NX]) const { M::rhs(kp, hold, x, dx); }
};
struct NoEp {  // EpWeights of a model without end-point groups
  void operator()(double, int (&kg)[2]) const { kg[0] = kg[1] = 0; }
  static constexpr int group(int) { return 0; }
};

static uint64_t digest = 1469598103934665603ull;
static void mix(const double* v, int n) {
  for (int i = 0; i < n; ++i) {
    uint64_t b; std::memcpy(&b, &v[i], 8);
    digest = (digest ^ b) * 1099511628211ull;
  }
}
static double urand(uint64_t& s) {  // splitmix64 -> [0, 1)
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <int ID, int INTEG>
static int check_model(const char* name, const double* raw, int nraw, const double* xlo, const double* xhi, const double* ulo,
                       const double* uhi, double dt) {
  using M = Model<ID>;
  constexpr int NX = M::NX, NU = M::NA + M::NDM;
  typename M::KP kp;
  double ddef[8] = {0};
  std::vector<double> rawv(raw, raw + nraw);
  M::prep(rawv.data(), NX, NU, reinterpret_cast<double*>(&kp), ddef);
  constexpr int T = ros_threads(NX);
  // the per-lane matrices exactly as the kernels size them (ros_lds_doubles): an index past the end is a heap overflow
  std::vector<double> lds(ros_lds_doubles(NX));
  uint64_t seed = 1234 + ID * 17 + INTEG;
  int bad = 0;
  for (int lane = 0; lane < 4 * T; ++lane) {
    threadIdx.x = lane % T;
    double x[NX], u[NU], f0[NX], xa[NX], ea[NX], xb[NX], eb[NX];
    for (int i = 0; i < NX; ++i) x[i] = xlo[i] + (xhi[i] - xlo[i]) * urand(seed);
    for (int i = 0; i < NU; ++i) u[i] = ulo[i] + (uhi[i] - ulo[i]) * urand(seed);
    const typename M::Hold hold = M::hold(kp, u);
    const HostRhs<M> f{kp, hold};
    const RosLds<NX> Lm(lds.data());
    const RosDense<NX, HostRhs<M>> ls{f, Lm, NX, 1e-6, 1e-8};
    f(x, f0);
    const double h = dt * std::pow(10.0, -3.0 * urand(seed));
    // one attempt, unrolled (ros_pair_try under PCG_ROS_ROLLED_ABOVE) against rolled, bit for bit
    const bool oka = ros_pair_try<INTEG, NX>(f, ls, x, f0, h, xa, ea);
    const bool okb = ros_try_rolled<INTEG, NX>(f, ls, x, f0, h, xb, eb);
    if (oka != okb || std::memcmp(xa, xb, sizeof xa) || std::memcmp(ea, eb, sizeof ea)) {
      if (bad++ < 3) std::printf("  %s lane %d: unrolled != rolled (x[2] %.17g vs %.17g)\n", name, lane, xa[2], xb[2]);
    }
    mix(xa, NX), mix(ea, NX);
    // the whole controller loop over one env step
    double xs[NX];
    for (int i = 0; i < NX; ++i) xs[i] = x[i];
    int nacc = 0, nrej = 0;
    const int st = ros_pair<INTEG, NX>(f, ls, NoEp{}, xs, NX, dt, 1e-6, 1e-8, 100000, nacc, nrej);
    if (st != 0) bad++;
    for (int i = 0; i < NX; ++i)
      if (!(std::fabs(xs[i]) < 1e300)) bad++;
    mix(xs, NX);
    const double cnt[2] = {(double)nacc, (double)nrej};
    mix(cnt, 2);
  }
  std::printf("%-16s NX %2d %s: %d lanes, %s\n", name, NX, INTEG == PCG_INT_RODAS5 ? "rodas5" : "rodas4", 4 * T,
              bad ? "MISMATCH / FAILURE" : "unrolled == rolled bit for bit, every env step finite");
  return bad;
}

int main() {
  int bad = 0;
  {  // heat_exchanger: 24 states (model_classes.py:935-1044), the model of the round-5 bug
    const double raw[] = {1, 1, 1, 1, 2, 3, 1, 1, 1, 1, 1, 1};
    double xlo[24], xhi[24];
    for (int i = 0; i < 24; ++i) xlo[i] = 280.0, xhi[i] = 380.0;
    const double ulo[] = {0.1, 0.1, 350.0, 280.0}, uhi[] = {2.0, 2.0, 400.0, 320.0};
    bad += check_model<PCG_MODEL_HEAT_EX, PCG_INT_RODAS4>("heat_exchanger", raw, 12, xlo, xhi, ulo, uhi, 0.5);
    bad += check_model<PCG_MODEL_HEAT_EX, PCG_INT_RODAS5>("heat_exchanger", raw, 12, xlo, xhi, ulo, uhi, 0.5);
  }
  {  // biofilm_reactor: 16 states (model_classes.py:1046-1155) -- on the unrolled side of the threshold in the product
    const double raw[] = {10.0, 15.0, 1.5, 0.5, 1.0, 300, 0.8, 1.0, 0.5, 0.1, 1.5, 0.5};
    double xlo[16], xhi[16];
    for (int i = 0; i < 16; ++i) xlo[i] = 0.5, xhi[i] = 10.0;
    const double ulo[] = {1.0, 1.0, 5.0, 0.0, 0.0}, uhi[] = {5.0, 5.0, 20.0, 1.0, 1.0};
    bad += check_model<PCG_MODEL_BIOFILM, PCG_INT_RODAS4>("biofilm_reactor", raw, 12, xlo, xhi, ulo, uhi, 0.1);
    bad += check_model<PCG_MODEL_BIOFILM, PCG_INT_RODAS5>("biofilm_reactor", raw, 12, xlo, xhi, ulo, uhi, 0.1);
  }
  {  // the 20-state reactive cascade (model_classes.py:763-861), pow() form
    const double raw[] = {5.0, 5.0, 1.0, 0.01, 0.1, 2.0, 2.00, 0.00, 2.00, 0.00};
    double xlo[20], xhi[20];
    for (int i = 0; i < 20; ++i) xlo[i] = 0.05, xhi[i] = 1.5;
    const double ulo[] = {5.0, 10.0}, uhi[] = {50.0, 100.0};
    bad += check_model<PCG_MODEL_ME_REACTIVE, PCG_INT_RODAS4>("me_reactive", raw, 10, xlo, xhi, ulo, uhi, 1.0);
    bad += check_model<PCG_MODEL_ME_REACTIVE, PCG_INT_RODAS5>("me_reactive", raw, 10, xlo, xhi, ulo, uhi, 1.0);
  }
  std::printf("digest %016llx   (equal between the -ftrivial-auto-var-init=pattern and =zero builds: no result depends on an "
              "uninitialised local)\n", (unsigned long long)digest);
  return bad ? 1 : 0;
}
