// Poisson and normal variates from Philox streams.
//
// The reference draws these from numpy's legacy generator
// (np.random.poisson: exposure_generator.py:495,626, cosmic_rays.py:127;
//  np.random.normal: exposure_generator.py:328-329,725, detector.py:191,198).
// numpy's legacy poisson is Hoermann's PTRS transformed rejection for
// lam >= 10 and Knuth's product-of-uniforms below; the same two published
// algorithms are used here so the distribution is exactly Poisson, but fed
// from Philox words.  Everything is written without fused multiply-adds
// (the library is built with -ffp-contract=off) so the CPU oracle, which
// restates these formulas independently, sees the same roundings.
#pragma once
#include <math.h>
#include "philox.h"

namespace wayne {

template <class T> struct fp;
template <> struct fp<float> {
  static WAYNE_HD float u01(uint32_t x) { return u01f(x); }
  static WAYNE_HD float log_(float x) { return logf(x); }
  static WAYNE_HD float exp_(float x) { return expf(x); }
  static WAYNE_HD float sqrt_(float x) { return sqrtf(x); }
  static WAYNE_HD float floor_(float x) { return floorf(x); }
  static WAYNE_HD float abs_(float x) { return fabsf(x); }
};
template <> struct fp<double> {
  static WAYNE_HD double u01(uint32_t x) { return u01d(x); }
  static WAYNE_HD double log_(double x) { return log(x); }
  static WAYNE_HD double exp_(double x) { return exp(x); }
  static WAYNE_HD double sqrt_(double x) { return sqrt(x); }
  static WAYNE_HD double floor_(double x) { return floor(x); }
  static WAYNE_HD double abs_(double x) { return fabs(x); }
};

// ln Gamma(x), x >= 1: shift to x >= 7 then the Stirling series.
template <class T>
WAYNE_HD T loggam(T x) {
  T prod = (T)1;
  for (int i = 0; i < 6 && x < (T)7; ++i) {
    prod = prod * x;
    x = x + (T)1;
  }
  const T xi = (T)1 / x;
  const T x2 = xi * xi;
  T s = (T)(-691.0 / 360360.0);
  s = s * x2 + (T)(1.0 / 1188.0);
  s = s * x2 + (T)(-1.0 / 1680.0);
  s = s * x2 + (T)(1.0 / 1260.0);
  s = s * x2 + (T)(-1.0 / 360.0);
  s = s * x2 + (T)(1.0 / 12.0);
  s = s * xi;
  return (x - (T)0.5) * fp<T>::log_(x) - x + (T)0.91893853320467274178 + s -
         fp<T>::log_(prod);
}

// Poisson(lam).  Returned as T (an integer value); lam <= 0 gives 0.
// Every loop has a hard iteration cap so that no wave can spin forever.
template <class T>
WAYNE_HD T poisson(T lam, PhiloxStream& rng) {
  if (!(lam > (T)0)) return (T)0;
  if (lam < (T)10) {
    const T enlam = fp<T>::exp_(-lam);
    T k = (T)0;
    T prod = (T)1;
    for (int it = 0; it < 4096; ++it) {
      prod = prod * fp<T>::u01(rng.next());
      if (prod > enlam)
        k = k + (T)1;
      else
        break;
    }
    return k;
  }
  const T slam = fp<T>::sqrt_(lam);
  const T loglam = fp<T>::log_(lam);
  const T b = (T)0.931 + (T)2.53 * slam;
  const T a = (T)-0.059 + (T)0.02483 * b;
  const T invalpha = (T)1.1239 + (T)1.1328 / (b - (T)3.4);
  const T vr = (T)0.9277 - (T)3.6224 / (b - (T)2);
  for (int it = 0; it < 256; ++it) {
    const T U = fp<T>::u01(rng.next()) - (T)0.5;
    const T V = fp<T>::u01(rng.next());
    const T us = (T)0.5 - fp<T>::abs_(U);
    const T k = fp<T>::floor_(((T)2 * a / us + b) * U + lam + (T)0.43);
    if (us >= (T)0.07 && V <= vr) return k;
    if (k < (T)0 || (us < (T)0.013 && V > us)) continue;
    const T lhs = fp<T>::log_(V) + fp<T>::log_(invalpha) - fp<T>::log_(a / (us * us) + b);
    const T rhs = -lam + k * loglam - loggam<T>(k + (T)1);
    if (lhs <= rhs) return k;
  }
  return fp<T>::floor_(lam + (T)0.5);
}

}  // namespace wayne
