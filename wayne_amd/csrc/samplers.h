// Poisson and normal variates from Philox / Philox-seeded streams.
//
// The reference draws these from numpy's legacy generator
// (np.random.poisson: exposure_generator.py:495,626, cosmic_rays.py:127;
//  np.random.normal: exposure_generator.py:328-329,725, detector.py:191,198).
// numpy's legacy poisson is Hoermann's PTRS transformed rejection for
// lam >= 10 and Knuth's product-of-uniforms below; the same two published
// algorithms are used here so the distribution is exactly Poisson, fed from
// counter-keyed streams.
//
// Two math policies:
//   ExactMath<T>  IEEE divide / sqrt and the ocml log / exp, no fused
//                 multiply-adds (the library is built with -ffp-contract=off):
//                 the CPU oracle, which restates these formulas independently,
//                 sees the same roundings except for 1-ulp libm differences.
//   FastMath      float only: v_rcp_f32 / v_sqrt_f32 / v_log_f32 / v_exp_f32.
//                 Same algorithm, same uniforms; a decision can differ from
//                 ExactMath only when it falls within ~1e-6 of a boundary.
#pragma once
#include <math.h>
#include "philox.h"

namespace wayne {

template <class T> struct ExactMath;
template <> struct ExactMath<float> {
  typedef float type;
  static WAYNE_HD float u01(uint32_t x) { return u01f(x); }
  static WAYNE_HD float log_(float x) { return logf(x); }
  static WAYNE_HD float exp_(float x) { return expf(x); }
  static WAYNE_HD float sqrt_(float x) { return sqrtf(x); }
  static WAYNE_HD float floor_(float x) { return floorf(x); }
  static WAYNE_HD float abs_(float x) { return fabsf(x); }
  static WAYNE_HD float div_(float a, float b) { return a / b; }
  // PTRS acceptance: log V + log(1/alpha) - log(a/us^2 + b) <= rhs
  static WAYNE_HD bool accept(float V, float invalpha, float den, float rhs) {
    return logf(V) + logf(invalpha) - logf(den) <= rhs;
  }
};
template <> struct ExactMath<double> {
  typedef double type;
  static WAYNE_HD double u01(uint32_t x) { return u01d(x); }
  static WAYNE_HD double log_(double x) { return log(x); }
  static WAYNE_HD double exp_(double x) { return exp(x); }
  static WAYNE_HD double sqrt_(double x) { return sqrt(x); }
  static WAYNE_HD double floor_(double x) { return floor(x); }
  static WAYNE_HD double abs_(double x) { return fabs(x); }
  static WAYNE_HD double div_(double a, double b) { return a / b; }
  static WAYNE_HD bool accept(double V, double invalpha, double den, double rhs) {
    return log(V) + log(invalpha) - log(den) <= rhs;
  }
};

#if defined(__HIP_DEVICE_COMPILE__)
struct FastMath {
  typedef float type;
  static __device__ __forceinline__ float u01(uint32_t x) { return u01f(x); }
  static __device__ __forceinline__ float log_(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
  static __device__ __forceinline__ float exp_(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
  static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float floor_(float x) { return floorf(x); }
  static __device__ __forceinline__ float abs_(float x) { return fabsf(x); }
  static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  // the three logs folded into one: log(V * invalpha / den)
  static __device__ __forceinline__ bool accept(float V, float invalpha, float den, float rhs) {
    return log_(V * invalpha * __builtin_amdgcn_rcpf(den)) <= rhs;
  }
};
#else
typedef ExactMath<float> FastMath;   // host pass: never executed
#endif

// ln Gamma(x), x >= 1: shift to x >= 7 then the Stirling series.
template <class M>
WAYNE_HD typename M::type loggam(typename M::type x) {
  typedef typename M::type T;
  T prod = (T)1;
  for (int i = 0; i < 6 && x < (T)7; ++i) {
    prod = prod * x;
    x = x + (T)1;
  }
  const T xi = M::div_((T)1, x);
  const T x2 = xi * xi;
  T s = (T)(-691.0 / 360360.0);
  s = s * x2 + (T)(1.0 / 1188.0);
  s = s * x2 + (T)(-1.0 / 1680.0);
  s = s * x2 + (T)(1.0 / 1260.0);
  s = s * x2 + (T)(-1.0 / 360.0);
  s = s * x2 + (T)(1.0 / 12.0);
  s = s * xi;
  return (x - (T)0.5) * M::log_(x) - x + (T)0.91893853320467274178 + s - M::log_(prod);
}

// The constants of one PTRS draw (Hoermann 1993, algorithm PTRS).
template <class M>
struct PtrsSetup {
  typedef typename M::type T;
  T lam, b, a, vr, loglam, invalpha;
  WAYNE_HD void init(T lam_) {
    lam = lam_;
    const T slam = M::sqrt_(lam);
    loglam = M::log_(lam);
    b = (T)0.931 + (T)2.53 * slam;
    a = (T)-0.059 + (T)0.02483 * b;
    invalpha = (T)1.1239 + M::div_((T)1.1328, b - (T)3.4);
    vr = (T)0.9277 - M::div_((T)3.6224, b - (T)2);
  }
  // One trial with uniforms (u1, u2): true and k set when accepted.
  WAYNE_HD bool trial(uint32_t w1, uint32_t w2, T& k) const {
    const T U = M::u01(w1) - (T)0.5;
    const T V = M::u01(w2);
    const T us = (T)0.5 - M::abs_(U);
    k = M::floor_((M::div_((T)2 * a, us) + b) * U + lam + (T)0.43);
    if (us >= (T)0.07 && V <= vr) return true;
    if (k < (T)0 || (us < (T)0.013 && V > us)) return false;
    return M::accept(V, invalpha, M::div_(a, us * us) + b, -lam + k * loglam - loggam<M>(k + (T)1));
  }
};

// Poisson(lam) from any stream with next().  Returned as T (an integer
// value); lam <= 0 gives 0.  Every loop has a hard iteration cap so that no
// wave can spin forever.
template <class M, class RNG>
WAYNE_HD typename M::type poisson(typename M::type lam, RNG& rng) {
  typedef typename M::type T;
  if (!(lam > (T)0)) return (T)0;
  if (lam < (T)10) {
    const T enlam = M::exp_(-lam);
    T k = (T)0;
    T prod = (T)1;
    for (int it = 0; it < 4096; ++it) {
      prod = prod * M::u01(rng.next());
      if (prod > enlam)
        k = k + (T)1;
      else
        break;
    }
    return k;
  }
  PtrsSetup<M> ps;
  ps.init(lam);
  for (int it = 0; it < 256; ++it) {
    const uint32_t w1 = rng.next();
    const uint32_t w2 = rng.next();
    T k;
    if (ps.trial(w1, w2, k)) return k;
  }
  return M::floor_(lam + (T)0.5);
}

}  // namespace wayne
