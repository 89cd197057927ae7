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
  static constexpr bool fast = false;
  static WAYNE_HD float u01(uint32_t x) { return u01f(x); }
  static WAYNE_HD float log_(float x) { return logf(x); }
  static WAYNE_HD float exp_(float x) { return expf(x); }
  static WAYNE_HD float sqrt_(float x) { return sqrtf(x); }
  static WAYNE_HD float floor_(float x) { return floorf(x); }
  static WAYNE_HD float abs_(float x) { return fabsf(x); }
  static WAYNE_HD float div_(float a, float b) { return a / b; }
  static WAYNE_HD float log1m_(float p) { return log1pf(-p); }   // ln(1 - p) without losing a small p
  // PTRS acceptance: log V + log(1/alpha) - log(a/us^2 + b) <= rhs
  static WAYNE_HD bool accept(float V, float invalpha, float den, float rhs) {
    return logf(V) + logf(invalpha) - logf(den) <= rhs;
  }
};
template <> struct ExactMath<double> {
  typedef double type;
  static constexpr bool fast = false;
  static WAYNE_HD double u01(uint32_t x) { return u01d(x); }
  static WAYNE_HD double log_(double x) { return log(x); }
  static WAYNE_HD double exp_(double x) { return exp(x); }
  static WAYNE_HD double sqrt_(double x) { return sqrt(x); }
  static WAYNE_HD double floor_(double x) { return floor(x); }
  static WAYNE_HD double abs_(double x) { return fabs(x); }
  static WAYNE_HD double div_(double a, double b) { return a / b; }
  static WAYNE_HD double log1m_(double p) { return log1p(-p); }
  static WAYNE_HD bool accept(double V, double invalpha, double den, double rhs) {
    return log(V) + log(invalpha) - log(den) <= rhs;
  }
};

#if defined(__HIP_DEVICE_COMPILE__)
struct FastMath {
  typedef float type;
  static constexpr bool fast = true;
  static __device__ __forceinline__ float u01(uint32_t x) { return u01f(x); }
  static __device__ __forceinline__ float log_(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
  static __device__ __forceinline__ float exp_(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
  static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float floor_(float x) { return floorf(x); }
  static __device__ __forceinline__ float abs_(float x) { return fabsf(x); }
  static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
  static __device__ __forceinline__ float log1m_(float p) {
    return p < 0.01f ? -p * (1.f + p * (0.5f + p * (0.33333334f + 0.25f * p))) : log_(1.f - p);
  }
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
  // The cheap part of a trial with uniforms (w1, w2): 1 = accepted (k set), 0 = rejected,
  // 2 = undecided -- call slow(k, V, us).  ~87 % of trials end here.
  WAYNE_HD int quick(uint32_t w1, uint32_t w2, T& k, T& V, T& us) const {
    const T U = M::u01(w1) - (T)0.5;
    V = M::u01(w2);
    us = (T)0.5 - M::abs_(U);
    k = M::floor_((M::div_((T)2 * a, us) + b) * U + lam + (T)0.43);
    if (us >= (T)0.07 && V <= vr) return 1;
    if (k < (T)0 || (us < (T)0.013 && V > us)) return 0;
    return 2;
  }
  // The log-density comparison of the undecided trials.
  WAYNE_HD bool slow(T k, T V, T us) const {
    return M::accept(V, invalpha, M::div_(a, us * us) + b, -lam + k * loglam - loggam<M>(k + (T)1));
  }
  // One whole trial: true and k set when accepted.
  WAYNE_HD bool trial(uint32_t w1, uint32_t w2, T& k) const {
    T V, us;
    const int q = quick(w1, w2, k, V, us);
    return q == 1 || (q == 2 && slow(k, V, us));
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
    uint32_t w1, w2;
    rng.next2(w1, w2);
    T k;
    if (ps.trial(w1, w2, k)) return k;
  }
  return M::floor_(lam + (T)0.5);
}

// ---------------------------------------------------------------------------
// Binomial(n, p): exact.  Inversion by sequential search (Kachitvichyanukul &
// Schmeiser's BINV) when n min(p, 1-p) < 10, Hoermann's BTRS transformed
// rejection otherwise ("The generation of binomial random variates", J. Stat.
// Comput. Simul. 46, 1993) -- the binomial sibling of the PTRS sampler above.
// Used to split the electrons of a bin over pixels as ONE multinomial draw
// instead of throwing them one by one (k_narrow).
// ---------------------------------------------------------------------------
// ln k! - [ln sqrt(2 pi) + (k + 1/2) ln(k + 1) - (k + 1)]
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const double kStirlingSmall[10] = {
#else
static const double kStirlingSmall[10] = {
#endif
    0.0810614667953272,  0.0413406959554092, 0.0276779256849983, 0.02079067210376509, 0.0166446911898211,
    0.0138761288230707,  0.0118967099458917, 0.0104112652619720, 0.00925546218271273, 0.00833056343336287};

// `small`: the table of the ten exact values in the caller's fastest memory (k_narrow keeps a float copy in LDS:
// the lanes index it with different k, which from the constant array would be a global load per trial)
template <class M>
WAYNE_HD typename M::type stirling_tail(typename M::type k, const float* small = nullptr) {
  typedef typename M::type T;
  if (k < (T)10) return small ? (T)small[(int)k] : (T)kStirlingSmall[(int)k];   // exact values for k = 0..9
  const T kp1 = k + (T)1;
  if (M::fast) {   // one reciprocal instead of three divisions
    const T r1 = M::div_((T)1, kp1), r2 = r1 * r1;
    return r1 * ((T)(1.0 / 12) - r2 * ((T)(1.0 / 360) - r2 * (T)(1.0 / 1260)));
  }
  const T kp1sq = kp1 * kp1;
  return M::div_((T)(1.0 / 12) - M::div_((T)(1.0 / 360) - M::div_((T)(1.0 / 1260), kp1sq), kp1sq), kp1);
}

// Inversion below this mean, transformed rejection above (BTRS is valid from a mean of 10).  Measured on cfg4:
// switching at 30 instead makes k_narrow 15 % slower -- a wave pays for its longest search.
constexpr int kBinvMax = 10;

template <class M, class RNG>
WAYNE_HD typename M::type binomial(typename M::type n, typename M::type p, RNG& rng, const float* small = nullptr) {
  typedef typename M::type T;
  if (!(n > (T)0) || !(p > (T)0)) return (T)0;
  if (p >= (T)1) return n;
  const bool flip = p > (T)0.5;
  if (flip) p = (T)1 - p;
  const T q = (T)1 - p;
  T x = (T)0;
  if (n * p < (T)kBinvMax) {
    // BINV: walk the pmf from 0 with f(x) = f(x-1) ((n+1) s / x - s), s = p/q
    const T s = M::div_(p, q);
    const T a = (n + (T)1) * s;
    T r = M::exp_(n * M::log1m_(p));
    T u = M::u01(rng.next());
    // (n p < 10: a legitimate draw passes 64 with probability < 1e-25.  A search that gets there has a uniform that
    // fell into the rounding residue of the pmf's float sum -- ~6e-8 of the draws -- and would walk on for ever: it
    // takes the mean instead.  A per-step test for that costs the kernel 2.5 %; the cap costs nothing.)
    for (int it = 0; it < 64; ++it) {
      if (u <= r) break;
      u = u - r;
      x = x + (T)1;
      r = r * (M::div_(a, x) - s);
      if (x >= n) { x = n; break; }
    }
    if (x >= (T)64 && x < n) x = M::floor_(n * p + (T)0.5);
  } else {
    const T spq = M::sqrt_(n * p * q);
    const T b = (T)1.15 + (T)2.53 * spq;
    const T a = (T)-0.0873 + (T)0.0248 * b + (T)0.01 * p;
    const T c = n * p + (T)0.5;
    const T vr = (T)0.92 - M::div_((T)4.2, b);
    const T alpha = ((T)2.83 + M::div_((T)5.1, b)) * spq;
    const T m = M::floor_((n + (T)1) * p);
    const T r = M::div_(p, q);
    // the terms of the acceptance bound that do not depend on the trial
    const T nm = n - m + (T)1;
    const T bound_m = (m + (T)0.5) * M::log_(M::div_(m + (T)1, r * nm)) + stirling_tail<M>(m, small) + stirling_tail<M>(n - m, small);
    x = M::floor_(n * p + (T)0.5);   // returned only if the (unreachable) iteration cap is hit
    for (int it = 0; it < 256; ++it) {
      uint32_t wu, wv;
      rng.next2(wu, wv);
      const T U = M::u01(wu) - (T)0.5;
      const T V = M::u01(wv);
      const T us = (T)0.5 - M::abs_(U);
      const T k = M::floor_((M::div_((T)2 * a, us) + b) * U + c);
      if (us >= (T)0.07 && V <= vr) { x = k; break; }
      if (k < (T)0 || k > n) continue;
      const T v = M::log_(M::div_(V * alpha, M::div_(a, us * us) + b));
      const T nk = n - k + (T)1;
      const T ub = bound_m + (n + (T)1) * M::log_(M::div_(nm, nk)) +
                   (k + (T)0.5) * M::log_(M::div_(r * nk, k + (T)1)) - stirling_tail<M>(k, small) - stirling_tail<M>(n - k, small);
      if (v <= ub) { x = k; break; }
    }
  }
  return flip ? n - x : x;
}

}  // namespace wayne
