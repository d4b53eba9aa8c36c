// pcg_pack.hpp -- W environments per lane as one value type.
//
// A wave that integrates ONE env per lane executes a single chain of dependent fp64 instructions
// (RK stage -> exp -> divide -> ...), and a dependent fp64 op can only issue every ~8-10 cycles:
// tools/overlapbench.hip measures 36.7 us for 256 dependent FMAs per env against 19.0 us when the
// same work is two independent chains.  Pack<W> carries W envs through the same code, so every
// operator becomes W independent instructions the scheduler interleaves: the VALU issues
// back-to-back, a wave finishes its arithmetic sooner, and memory accesses are 8*W bytes per lane.
//
// Pack<1> is a plain double in a struct: one code path for both widths.
#pragma once
#include <hip/hip_runtime.h>

namespace pcg {

#define PCG_PK __device__ __forceinline__

template <int W>
struct Pack {
  double v[W];
  PCG_PK Pack() {}
  PCG_PK Pack(double s) {
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = s;
  }
};

#define PCG_PACK_BINOP(OP)                                                  \
  template <int W>                                                          \
  PCG_PK Pack<W> operator OP(const Pack<W>& a, const Pack<W>& b) {          \
    Pack<W> r;                                                              \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = a.v[i] OP b.v[i]; \
    return r;                                                               \
  }                                                                         \
  template <int W>                                                          \
  PCG_PK Pack<W> operator OP(const Pack<W>& a, double b) {                  \
    Pack<W> r;                                                              \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = a.v[i] OP b;     \
    return r;                                                               \
  }                                                                         \
  template <int W>                                                          \
  PCG_PK Pack<W> operator OP(double a, const Pack<W>& b) {                  \
    Pack<W> r;                                                              \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = a OP b.v[i];     \
    return r;                                                               \
  }
PCG_PACK_BINOP(+)
PCG_PACK_BINOP(-)
PCG_PACK_BINOP(*)
PCG_PACK_BINOP(/)
#undef PCG_PACK_BINOP

template <int W>
PCG_PK Pack<W> operator-(const Pack<W>& a) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = -a.v[i];
  return r;
}
template <int W>
PCG_PK Pack<W>& operator+=(Pack<W>& a, const Pack<W>& b) {
#pragma unroll
  for (int i = 0; i < W; ++i) a.v[i] += b.v[i];
  return a;
}

#define PCG_PACK_FN1(NAME)                                           \
  template <int W>                                                   \
  PCG_PK Pack<W> NAME(const Pack<W>& a) {                            \
    Pack<W> r;                                                       \
    _Pragma("unroll") for (int i = 0; i < W; ++i) r.v[i] = ::NAME(a.v[i]); \
    return r;                                                        \
  }
PCG_PACK_FN1(exp)
PCG_PACK_FN1(log)
PCG_PACK_FN1(sqrt)
PCG_PACK_FN1(fabs)
#undef PCG_PACK_FN1

template <int W>
PCG_PK Pack<W> pow(const Pack<W>& a, double e) {
  Pack<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = ::pow(a.v[i], e);
  return r;
}

// scalar overloads so model code can be written once for T = double and T = Pack<W>
PCG_PK double pk_get(double a, int) { return a; }
template <int W>
PCG_PK double pk_get(const Pack<W>& a, int i) {
  return a.v[i];
}

}  // namespace pcg
