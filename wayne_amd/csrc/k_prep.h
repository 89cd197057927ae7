// k_prep_wl, k_prep_sub, k_prep_fix: per-wavelength arrays, per-bin positions and counts (A6-A10); cosmic-ray hits (A13)
#pragma once
#include "common.h"

namespace wayne {

// ---------------------------------------------------------------------------
// k_prep_wl : A8 + the wavelength-only part of A9
// ---------------------------------------------------------------------------
// (trace_coeffs: plan_consts.h -- the host's accumulator boxes use the same function)

// (also: thread k computes the trace coefficients of sub-sample k -- six numbers that every workgroup of k_prep_sub
// needs before it can start; computed there by one thread with the other 511 waiting at a barrier they were ~2 us of
// serial fp64 latency per workgroup)
__global__ void k_prep_wl(GrismDev g, int W, const double* __restrict__ wl, WlArrays o, uint32_t* misc, int K,
                          const double* __restrict__ x_ref, const double* __restrict__ y_ref, double* __restrict__ tr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the exposure's status words (total electrons, overflow flag) start from zero: cleared here, by the
  // first kernel of the exposure, instead of by a separate fill in front of it
  if (i < 16) misc[i] = 0u;
  if (i < K) {
    double* t = tr + kTrStride * (size_t)i;
    trace_coeffs(g, x_ref[i], y_ref[i], t);
    t[6] = 1. / t[4];     // (for bin_position_bound)
    t[7] = 0.;
  }
  if (i >= W) return;
  const double x = wl[i];
  o.ratio[i] = poly3(g.p_ratio, x);
  o.sigl[i] = poly3(g.p_sigl, x);
  o.sigh[i] = poly3(g.p_sigh, x);

  // np.interp (grism.py:116-118): clamp outside the table, linear inside.
  double s;
  const int n = g.n_sens;
  if (n <= 0) {
    s = 1.0;
  } else if (n < 2 || x <= g.sens_wl[0]) {      // (a one-entry table has no bracket to search: its value, NaN for a NaN wavelength)
    s = (x != x) ? x : g.sens_val[0];
  } else if (x >= g.sens_wl[n - 1]) {
    s = g.sens_val[n - 1];
  } else {
    // sens_wl[lo] <= x < sens_wl[lo + 1].  The tables are (nearly) uniform in wavelength: a proportional guess is
    // right or off by one, two dependent loads instead of the log2(n) of a bisection (the kernel is little but this
    // latency: 8.3 -> 6.9 us); where the guess fails -- a table with gaps -- the bisection runs over what is left
    const double w0 = g.sens_wl[0], w1 = g.sens_wl[n - 1];
    int lo = (int)((x - w0) / (w1 - w0) * (double)(n - 1));
    lo = min(max(lo, 0), n - 2);
    if (g.sens_wl[lo] > x) {
      if (lo > 0 && g.sens_wl[lo - 1] <= x) {
        lo -= 1;
      } else {
        int hi = lo;
        lo = 0;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (g.sens_wl[mid] <= x) lo = mid; else hi = mid; }
      }
    } else if (!(x < g.sens_wl[lo + 1])) {
      if (lo + 2 <= n - 1 && x < g.sens_wl[lo + 2]) {
        lo += 1;
      } else {
        int hi = n - 1;
        lo += 1;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (g.sens_wl[mid] <= x) lo = mid; else hi = mid; }
      }
    }
    lo = min(lo, n - 2);     // (a NaN wavelength fails every comparison)
    const double slope = (g.sens_val[lo + 1] - g.sens_val[lo]) / (g.sens_wl[lo + 1] - g.sens_wl[lo]);
    s = slope * (x - g.sens_wl[lo]) + g.sens_val[lo];
  }
  o.sens[i] = s;

  // tools.bin_centers_to_widths (tools.py:106-128): half-gaps to both
  // neighbours; the end bins mirror their single neighbour.
  double left, right;
  if (W < 2) {
    left = right = 0.0;
  } else {
    left = (i == 0) ? (wl[1] - wl[0]) / 2. : (wl[i] - wl[i - 1]) / 2.;
    right = (i == W - 1) ? (wl[W - 1] - wl[W - 2]) / 2. : (wl[i + 1] - wl[i]) / 2.;
  }
  o.dlam[i] = left + right;
}

// ---------------------------------------------------------------------------
// k_prep_sub : one workgroup per sub-sample
// ---------------------------------------------------------------------------
struct PrepArgs {
  GrismDev g;
  int W, K, N;               // bins, sub-samples, light-sensitive side
  int sub_scale;             // 507 - SUBARRAY/2 (exposure_generator.py:630)
  int margin;                // LDS tile margin (px)
  int max_tile;              // LDS tile capacity (ints)
  uint32_t seed, exposure;
  uint32_t flags;
  int split_min;             // > 0: WAYNE_RNG_SPLIT -- bins with >= split_min narrow electrons go to k_narrow
  int lane_max;              // WAYNE_RNG_SPLIT: most one-by-one electrons a bin's own lane takes (kLaneMax; kLaneReach
                             // when the host launches no k_throw for the exposure because it expects no larger bin)
  double scale_factor;
  const double* wl;          // [W]
  const double* flux;        // [W]
  const double* depth;       // [K*W] or null
  const double* x_ref;       // [K]
  const double* y_ref;       // [K]
  const double* dur_ms;      // [K]
  const int32_t* replay_seed;  // [K]
  const int32_t* sample_read;  // [K]
  const double* tr;          // [K*6] trace coefficients per sub-sample (k_prep_wl): m_t, c_t, m_w, c_w, m_wl, c_wl
  WlArrays wa;
  // outputs
  int32_t* counts;           // [K*W]
  int32_t* nwide;            // [K*W]
  int32_t* nsplit;           // [K*W] split mode: narrow electrons handed to k_narrow's multinomial (else 0)
  int32_t* nlane;            // [K*W] split mode: electrons the bin's own lane throws one by one in k_lane (else 0)
  uint32_t* prefix;          // [K*(W+1)] exclusive prefix of the electrons k_throw throws one by one
  double* xpos;              // [K*W] frame coords (x_sub)
  double* ypos;              // [K*W]
  SubInfo* sub;              // [K]
  unsigned long long* total_electrons;  // += E_k
  int* status;               // set non-zero on overflow
  uint32_t* chunk_total;     // [K * n_chunks] electrons (for k_throw) per chunk of kPrepThreads bins
  double* chunk_box;         // [K * n_chunks * 4] xmin, xmax, ymin, ymax of the chunk's populated bins
  int fix_inline;            // split mode without a k_throw launch: chunk 0 writes SubInfo, k_prep_fix is not launched
  int no_narrow;             // k_narrow is not launched either (the host expects no bin with split_min narrow electrons)
};

// The device-side electron count (wayne_profile_get) is kept in kCounterStripes words a cache line apart, a workgroup adding
// to the stripe of its index: atomics on ONE address complete one after the other (8 ns apiece) and the 1152 workgroups of
// a k_prep_sub launch queued 2.7 us behind theirs.  The host adds the stripes up.
constexpr int kCounterStripes = 64;
constexpr int kCounterStride = 16;        // in 8-byte words: 128 B
__device__ __forceinline__ void count_electrons(unsigned long long* counter, unsigned wg, unsigned long long n) {
  atomicAdd(counter + (size_t)kCounterStride * (wg & (kCounterStripes - 1)), n);
}

#ifndef WAYNE_PREP_THREADS
#define WAYNE_PREP_THREADS 512
#endif
constexpr int kPrepThreads = WAYNE_PREP_THREADS;
constexpr int kMaxPrepChunks = 128;     // chunks of kPrepThreads bins per sub-sample (32768 bins)
// (kNarrowR, kLaneMax, kLaneReach, kSplitMaxNarrow, trace_coeffs: plan_consts.h)

// ---------------------------------------------------------------------------
// cosmic rays : MinMaxPossionCosmicGenerator.cosmic_frame (cosmic_rays.py:70-139)
// ---------------------------------------------------------------------------
// One workgroup per read interval adds the interval's hits to the accumulators (in electrons, before the
// gain: exposure_generator.py:497-505).  Fifteen small workgroups' worth of work: the first R workgroups of
// k_prep_sub do it on their way in instead of a launch of its own (which costs more than the work: ~8 us on the
// exposure's critical path).
struct CosmicArgs {
  int R, N, S;
  uint32_t seed, exposure;
  double rate;               // hits per second per 1024^2 pixels; < 0: no cosmic rays
  const double* read_dt;     // [R]
  long long* acc;            // [R*S*S]
  uint32_t* seg;             // [ceil(S*S / 64)] bit r: read interval r has a hit among the segment's 64 accumulators
                             // (k_ramp loads the accumulators outside the spectrum's box only where a bit says so)
};

__device__ __forceinline__ void cosmic_hits(const CosmicArgs& a, int r, uint32_t* s_n) {
  if (threadIdx.x == 0) {
    // rate_size = rate / (1024*1024) * N*N ; Poisson(rate_size * time)  (:33-44, :121-127)
    const double rate_size = a.rate / (1024. * 1024.) * (double)((long long)a.N * a.N);
    PhiloxStream rng(a.seed, STAGE_CR_COUNT, 0u, (uint32_t)r, a.exposure);
    double n = poisson<ExactMath<double> >(rate_size * a.read_dt[r], rng);
    if (!(n >= 0.)) n = 0.;
    if (n > 1e7) n = 1e7;
    *s_n = (uint32_t)n;
  }
  __syncthreads();
  const uint32_t n = *s_n;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const u32x4 w = philox4x32_10(i, 0u, (uint32_t)r, a.exposure, a.seed, STAGE_CR_HIT);
    const uint32_t energy = 10000u + uint_below(w.v[0], 25000u);  // randint(10000, 35000)  (:134)
    const uint32_t y = uint_below(w.v[1], (uint32_t)a.N);          // randint(0, len(array))  (:80)
    const uint32_t x = uint_below(w.v[2], (uint32_t)a.N);          // randint(0, len(array[0])) (:81)
    const long long q = (long long)energy << kQBits;
    const size_t pix = (size_t)(y + kBorder) * a.S + (x + kBorder);
    atomicAdd((unsigned long long*)&a.acc[(size_t)r * a.S * a.S + pix], (unsigned long long)q);
    atomicOr(&a.seg[pix >> 6], 1u << r);
  }
}

// ---------------------------------------------------------------------------
// The stellar Poisson draw of a bin: PTRS in fp64, same trials and same outcomes as
// poisson<ExactMath<double>> (samplers.h; the oracle's sampler), with a SQUEEZE in front
// of the log-density comparison.  A quarter of the trials reach that comparison
// (three fp64 logarithms and ln Gamma: ~400 fp64 instructions, and some lane of a
// wave always does).  Its two sides are first evaluated in fp32, the right-hand side
// in a form without cancellation,
//     -lam + k ln lam - ln k!  =  -d^2/lam - k g(d/lam) - ln(2 pi k)/2 - 1/(12k) + 1/(360k^3),
//     d = k - lam (exact in fp64),  g(x) = ln(1+x) - x  (series, |x| <= 1/4),
// good to 1.4e-6 (1 + d^2/lam) against the fp64 value (measured over 1e6 undecided
// trials, lam = 10 .. 4e6; scripts/check_ptrs_squeeze.py).  Only when the two sides
// are closer than 2e-4 (1 + d^2/lam) -- a margin of 140 -- does the fp64 comparison
// run: one trial in a few thousand.  The decision is the fp64 one either way.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
// > 0: accept, < 0: reject, 0: too close to call in fp32
__device__ __forceinline__ int ptrs_squeeze(double lam, double k, double lhs_arg) {
  const double d = k - lam;
  if (!(k >= 10. && lam <= 4194304. && fabs(d) <= 0.25 * lam)) return 0;
  const float df = (float)d, kf = (float)k;
  const float rl = __builtin_amdgcn_rcpf((float)lam), rk = __builtin_amdgcn_rcpf(kf);
  const float x = df * rl;
  // (ln(1+x) - x) / x^2 = -1/2 + x/3 - x^2/4 + ... - x^12/14 + x^13/15
  float p = 1.f / 15.f;
  p = fmaf(p, x, -1.f / 14.f); p = fmaf(p, x, 1.f / 13.f); p = fmaf(p, x, -1.f / 12.f);
  p = fmaf(p, x, 1.f / 11.f);  p = fmaf(p, x, -1.f / 10.f); p = fmaf(p, x, 1.f / 9.f);
  p = fmaf(p, x, -1.f / 8.f);  p = fmaf(p, x, 1.f / 7.f);   p = fmaf(p, x, -1.f / 6.f);
  p = fmaf(p, x, 1.f / 5.f);   p = fmaf(p, x, -1.f / 4.f);  p = fmaf(p, x, 1.f / 3.f);
  p = fmaf(p, x, -0.5f);
  const float scale = fmaf(df * df, rl, 1.f);                  // 1 + d^2 / lam
  const float ln2 = 0.6931471805599453f;
  const float rhs = -(df * df) * rl - kf * (p * x * x) - 0.5f * ln2 * __builtin_amdgcn_logf(6.283185307179586f * kf) -
                    rk * (1.f / 12.f - rk * rk * (1.f / 360.f));
  const float D = ln2 * __builtin_amdgcn_logf((float)lhs_arg) - rhs;   // accept <=> D <= 0
  const float eps = 2e-4f * scale;
  return (D <= -eps) ? 1 : ((D >= eps) ? -1 : 0);
}

// ... and below a mean of 10: ONE uniform and a sequential search of the cdf from 0 (the oracle's wo_poisson_counts),
// in fp32 with a guard band and in fp64 only when the uniform falls within it (a few draws in ten thousand): the
// decision is the fp64 one either way.  (numpy's product of uniforms takes lam + 1 words -- a second Philox block for
// a fifth word, a third for a ninth, each paid by the whole wave -- an fp64 exp and an fp64 product per word; on a
// finely sampled scan, millions of such draws per exposure, that was most of k_prep_sub.)
template <class RNG>
__device__ __forceinline__ double poisson_small(double lam, RNG& rng) {
  const double u = u01d(rng.next());
  const float lf = (float)lam, uf = (float)u;
  float p = FastMath::exp_(-lf), c = p;
  int k = 0;
  bool unsure = true;
  for (int it = 1; it < 64; ++it) {
    const float tol = fmaf(c, 2e-5f, 2e-7f);              // fp32 exp, product and sum: < 3e-6 c; uf: 6e-8
    if (uf <= c - tol) { unsure = false; break; }
    if (!(uf > c + tol)) break;
    p = p * (lf * (1.0f / (float)it));
    c = c + p;
    k = it;
  }
  if (unsure) {
    double pd = exp(-lam), cd = pd;
    k = 0;
    for (int it = 1; it < 256; ++it) {
      if (u <= cd) break;
      pd = pd * lam / (double)it;
      cd = cd + pd;
      k = it;
    }
  }
  return (double)k;
}

template <class RNG>
__device__ double poisson_counts(double lam, RNG& rng) {
  typedef ExactMath<double> M;
  if (!(lam > 0.)) return 0.;                                  // (also NaN)
  if (!(lam >= 10.)) return poisson_small(lam, rng);
  // PtrsSetup<M>::init without ln(lam), which only the fp64 comparison needs
  const double slam = sqrt(lam);
  const double b = 0.931 + 2.53 * slam;
  const double a = -0.059 + 0.02483 * b;
  const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
  const double vr = 0.9277 - 3.6224 / (b - 2.);
  for (int it = 0; it < 256; ++it) {
    uint32_t w1, w2;
    rng.next2(w1, w2);
    const double U = M::u01(w1) - 0.5;
    const double V = M::u01(w2);
    const double us = 0.5 - fabs(U);
    const double k = floor((2. * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return k;
    if (k < 0. || (us < 0.013 && V > us)) continue;
    const double den = a / (us * us) + b;
    int s = ptrs_squeeze(lam, k, V * invalpha / den);
    if (s == 0) s = M::accept(V, invalpha, den, -lam + k * log(lam) - loggam<M>(k + 1.)) ? 1 : -1;
    if (s > 0) return k;
  }
  return floor(lam + 0.5);
}
#else
template <class RNG> __device__ double poisson_counts(double lam, RNG& rng) { return poisson<ExactMath<double> >(lam, rng); }   // host pass: never executed
#endif

// SubInfo of sub-sample k: what the throwers need of it (read interval, replay seed, the trace coefficients of the
// flat field), the electrons k_throw shares out and the LDS tile rectangle of its workgroups
__device__ __forceinline__ SubInfo make_sub_info(const PrepArgs& a, int k, double x_ref, double y_ref, const double* tr,
                                                 uint32_t E, double xmin, double xmax, double ymin, double ymax) {
  SubInfo si;
  si.electrons = E;
  si.read = a.sample_read[k];
  si.replay_seed = a.replay_seed ? a.replay_seed[k] : 0;
  si.pad_ = 0;
  si.x_ref = x_ref; si.y_ref = y_ref;
  si.a_t_i = 1. / tr[0];            // grism.py:367
  si.a_w = tr[2]; si.b_w = tr[3];
  si.inv_norm = 1. / sqrt(si.a_t_i * si.a_t_i + 1.);
  // LDS tile: bounding box of the populated trace + margin, clipped to the
  // frame's populated range [1, N) (pixel row / column 0 is never hit,
  // pyparallel_menu.c:93), shrunk symmetrically if it exceeds the LDS budget
  // (electrons outside the tile take the global-atomic path: speed only).
  int tx0 = 0, ty0 = 0, tw = 0, th = 0;
  if (E > 0 && xmax >= xmin) {
    int x0 = (int)floor(xmin) - a.margin, x1 = (int)floor(xmax) + a.margin + 1;
    int y0 = (int)floor(ymin) - a.margin, y1 = (int)floor(ymax) + a.margin + 1;
    x0 = max(x0, 1); y0 = max(y0, 1); x1 = min(x1, a.N); y1 = min(y1, a.N);
    if (x1 > x0 && y1 > y0) {
      tw = x1 - x0; th = y1 - y0;
      while ((long long)tw * th > a.max_tile && th > 1) { y0 += 1; th -= 2; if (th < 1) th = 1; }
      while ((long long)tw * th > a.max_tile && tw > 1) { x0 += 1; tw -= 2; if (tw < 1) tw = 1; }
      tx0 = x0; ty0 = y0;
    }
  }
  si.tx0 = tx0; si.ty0 = ty0; si.tw = tw; si.th = th;
  return si;
}

// What k_prep_sub works out for bin w of sub-sample k (A7, A9, A10 and the routing of the bin) -- also evaluated on the
// fly, from the same inputs by the same code, by the fused form of k_lane (thin exposures), which is why it is a
// function: position in the frame, electron count, the sigma split and who throws what.
struct BinPlan {
  double xs, ys;            // frame position (x_sub, y_sub)
  uint32_t count;           // electrons of the bin (A9)
  int32_t nwide;            // N = (int)(counts * psf_ratio) (pyparallel_menu.c:89), not yet clamped to the count
  uint32_t narrow;          // electrons handed to k_narrow's multinomial (0: none)
  uint32_t lane;            // electrons the bin's own lane throws one by one (0: none)
  uint32_t rest;            // electrons left for k_throw
  bool overflow;
};
// wl_to_x / wl_to_y (grism.py:651, 667-669), then the sub-array shift x_sub = x_pos - sub_scale
// (exposure_generator.py:630-632).  Product and sum are rounded separately, as numpy rounds them (never contracted
// into an fma, whatever the compiler's flags): k_prep_sub and the fused k_lane both evaluate this function and have to
// agree with each other, and with the oracle, to the bit.
__device__ __forceinline__ void bin_position(const PrepArgs& a, double wl, const double* tr, double x_ref, double y_ref,
                                             double* xs, double* ys) {
  const double m_t = tr[0], c_t = tr[1], m_wl = tr[4], c_wl = tr[5];
  const double x = (wl - c_wl) / m_wl;
  const double y = __dadd_rn(__dmul_rn(m_t, x - x_ref), c_t) + y_ref;
  *xs = x - (double)a.sub_scale;
  *ys = y - (double)a.sub_scale;
}
// The same position to ~1e-10 px without the division (tr[6] = 1 / m_wl): for bounding boxes, with a pixel to spare
__device__ __forceinline__ void bin_position_bound(const PrepArgs& a, double wl, const double* tr, double x_ref, double y_ref,
                                                   float* xs, float* ys) {
  const double x = (wl - tr[5]) * tr[6];
  const double y = fma(tr[0], x - x_ref, tr[1]) + y_ref;
  *xs = (float)(x - (double)a.sub_scale);
  *ys = (float)(y - (double)a.sub_scale);
}
// wl .. sigl_w: the bin's per-wavelength inputs; tr: the sub-sample's trace coefficients (k_prep_wl); depth: the
// transit depth of (k, w) or 0
__device__ __forceinline__ BinPlan plan_bin(const PrepArgs& a, int k, int w, double wl, double flux_w, double sens_w,
                                            double dlam_w, double ratio_w, double sigl_w, const double* tr,
                                            double x_ref, double y_ref, double dur, double depth) {
  BinPlan o;
  bin_position(a, wl, tr, x_ref, y_ref, &o.xs, &o.ys);
  // counts chain (exposure_generator.py:344-348, 602-628, 649-687):
  //   F (1 - depth) * Sens * dlam[um] * 1e4 [A/um] * dur[ms] * 1e-3 [s/ms] * scale
  double f = flux_w;
  if (a.depth) f = f * (1. - depth);
  double lam = f * sens_w;
  lam = lam * dlam_w;
  lam = lam * 1e4;
  lam = lam * dur;
  lam = lam * 1e-3;
  lam = lam * a.scale_factor;
  double cnt;
  if ((a.flags & (1u << 5)) != 0) {         // WAYNE_F_ADD_STELLAR_NOISE
    PhiloxStream rng(a.seed, STAGE_COUNTS, (uint32_t)w, (uint32_t)k, a.exposure);
    cnt = poisson_counts(lam, rng);         // np.random.poisson (:626)
  } else {
    cnt = rint(lam);                        // np.round, half to even (:628)
  }
  o.overflow = false;
  if (!(cnt >= 0.)) cnt = 0.;               // negative / NaN flux throws no electrons
  if (cnt > 2147483647.) { cnt = 2147483647.; o.overflow = true; }
  const uint32_t c = (uint32_t)cnt;
  o.count = c;
  // N = counts*psf_ratio truncated (pyparallel_menu.c:89), in fp64
  const double nw = (double)(int32_t)c * ratio_w;
  o.nwide = (nw >= 2147483647.) ? 2147483647 : (nw <= -2147483648.) ? (int32_t)(-2147483647 - 1) : (int32_t)nw;
  // WAYNE_RNG_SPLIT: the narrow component of a well-populated bin is drawn as one multinomial by k_narrow;
  // what is left to throw one by one -- its wide electrons, or the whole of a bin that does not qualify --
  // is thrown by the bin's own lane in k_lane (no prefix search, no bin changes inside a lane's loop);
  // only a bin with more than kLaneMax such electrons is shared out by k_throw
  o.narrow = 0u; o.lane = 0u; o.rest = c;
  if (a.nsplit) {
    const uint32_t wide = (uint32_t)min(max(o.nwide, 0), (int32_t)min(c, 0x7FFFFFFFu));
    const uint32_t narrow = c - wide;
    const bool split = a.split_min > 0 && narrow >= (uint32_t)a.split_min && narrow <= kSplitMaxNarrow && sigl_w > 0.05 &&
                       sigl_w * 6.5 <= (double)kNarrowR;
    const uint32_t ind = split ? wide : c;                 // electrons thrown one by one
    const bool lane = a.split_min > 0 && ind <= (uint32_t)a.lane_max;
    o.narrow = split ? narrow : 0u;
    o.lane = lane ? ind : 0u;
    o.rest = lane ? 0u : ind;
  }
  return o;
}

// One workgroup per (sub-sample, chunk of kPrepThreads bins): positions, counts,
// sigma split, and the chunk-local exclusive prefix; k_prep_fix then adds the chunk
// offsets (when anything is left for k_throw: see the end of the kernel).
// K * ceil(W / 512) workgroups instead of K: the whole chip works.
__global__ __launch_bounds__(kPrepThreads) void k_prep_sub(PrepArgs a, CosmicArgs ca) {
  const int k = blockIdx.x;
  const int ch = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int W = a.W;
  constexpr int NW = kPrepThreads / 64;
  __shared__ uint32_t s_wsum[NW];   // per-wave totals
  __shared__ double s_red[4][NW];
  __shared__ uint32_t s_hits;

  if (ca.rate >= 0.) {   // cosmic rays: read interval r is the business of workgroup r (mod the grid)
    const int n_wg = gridDim.x * gridDim.y;
    for (int r = blockIdx.y * gridDim.x + blockIdx.x; r < ca.R; r += n_wg) { cosmic_hits(ca, r, &s_hits); __syncthreads(); }
  }
  const double x_ref = a.x_ref[k], y_ref = a.y_ref[k];
  const double* s_tr = a.tr + kTrStride * (size_t)k;      // (wave-uniform: scalar loads)
  const double dur = a.dur_ms[k];

  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  bool overflow = false;
  uint32_t n_split = 0;                   // per thread: electrons handed to k_lane / k_narrow

  const int w = ch * kPrepThreads + tid;
  uint32_t c = 0;
  if (w < W) {
    const size_t kw = (size_t)k * W + w;
    const BinPlan b = plan_bin(a, k, w, a.wl[w], a.flux[w], a.wa.sens[w], a.wa.dlam[w], a.wa.ratio[w], a.wa.sigl[w], s_tr,
                               x_ref, y_ref, dur, a.depth ? a.depth[kw] : 0.);
    a.xpos[kw] = b.xs;
    a.ypos[kw] = b.ys;
    a.counts[kw] = (int32_t)b.count;
    a.nwide[kw] = b.nwide;
    overflow = b.overflow;
    c = b.count;
    if (c > 0) {
      xmin = fmin(xmin, b.xs); xmax = fmax(xmax, b.xs);
      ymin = fmin(ymin, b.ys); ymax = fmax(ymax, b.ys);
    }
    if (a.nsplit) {
      a.nsplit[kw] = (int32_t)b.narrow;
      a.nlane[kw] = (int32_t)b.lane;
      n_split = b.narrow + b.lane;
      c = b.rest;                                            // c: electrons left for k_throw
      // launched without k_throw (the host expected no bin beyond a lane's reach) and here is one after all:
      // tell the host, which runs the exposure again with k_throw (wayne_hip.hip, check_status)
      if (a.fix_inline && c > 0u) atomicOr(a.status, 2);
      if (a.no_narrow && b.narrow > 0u) atomicOr(a.status, 2);     // ... or without k_narrow, and here is a bin for it
    }
  }
  if (overflow) atomicOr(a.status, 1);
  // electrons handed to k_lane / k_narrow: one atomic per workgroup (DPP row sums per wave -- six rounds of 64-bit
  // shuffles through the LDS crossbar at the end of a wave that has nothing to hide them behind were 1.8 us of the
  // launch --, then LDS)
  const unsigned long long n_split_total = wave_sum_u32(n_split);
  __shared__ unsigned long long s_split[NW];
  if (lane == 0) s_split[wave] = n_split_total;

  if (!a.fix_inline) {
    // what k_throw needs: exclusive scan of c inside the chunk (shuffle scan per wave, wave totals through LDS) ...
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    // ... and the bounding box of the populated bins
    for (int off = 32; off > 0; off >>= 1) {
      xmin = fmin(xmin, __shfl_down(xmin, off));
      xmax = fmax(xmax, __shfl_down(xmax, off));
      ymin = fmin(ymin, __shfl_down(ymin, off));
      ymax = fmax(ymax, __shfl_down(ymax, off));
    }
    if (lane == 0) {
      s_red[0][wave] = xmin; s_red[1][wave] = xmax;
      s_red[2][wave] = ymin; s_red[3][wave] = ymax;
    }
    __syncthreads();
    uint64_t wave_off = 0, chunk_total = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const uint32_t t = s_wsum[i];
      if (i < wave) wave_off += t;
      chunk_total += t;
    }
    if (chunk_total > 0xFFFFFFFFull) atomicOr(a.status, 1);
    if (w < W) a.prefix[(size_t)k * (W + 1) + w] = (uint32_t)(wave_off + incl - c);   // chunk-local for now
    if (tid == 0) {
      for (int i = 1; i < NW; ++i) {
        xmin = fmin(xmin, s_red[0][i]); xmax = fmax(xmax, s_red[1][i]);
        ymin = fmin(ymin, s_red[2][i]); ymax = fmax(ymax, s_red[3][i]);
      }
      const size_t ci = (size_t)k * gridDim.y + ch;
      a.chunk_total[ci] = (uint32_t)chunk_total;
      a.chunk_box[4 * ci + 0] = xmin; a.chunk_box[4 * ci + 1] = xmax;
      a.chunk_box[4 * ci + 2] = ymin; a.chunk_box[4 * ci + 3] = ymax;
    }
  }
  __syncthreads();
  if (tid == 0) {
    unsigned long long tot = 0;
    for (int i = 0; i < NW; ++i) tot += s_split[i];
    if (tot) count_electrons(a.total_electrons, blockIdx.y * gridDim.x + blockIdx.x, tot);
  }

  // Split mode with no k_throw launch (the default: every bin's one-by-one electrons fit its lane): nothing of the
  // sub-sample's SubInfo depends on the other workgroups -- no electrons for k_throw, no tile -- so chunk 0 writes it
  // here and k_prep_fix is not launched (~12 us of the exposure's critical path for ~1 us of work).  (A ticket
  // counter electing the last workgroup would serve every mode, but what the others wrote is only visible across
  // XCDs after an L2 write-back per workgroup: measured 0.17 ms per launch.)
  if (a.fix_inline && ch == 0 && tid == 0) a.sub[k] = make_sub_info(a, k, x_ref, y_ref, s_tr, 0u, 1e300, -1e300, 1e300, -1e300);
}

// One workgroup per sub-sample (modes that launch k_throw): chunk offsets -> global exclusive prefix, E_k,
// bounding box -> LDS tile rectangle, SubInfo.
__global__ __launch_bounds__(kPrepThreads) void k_prep_fix(PrepArgs a, int n_chunks) {
  const int k = blockIdx.x;
  const int tid = threadIdx.x;
  const int W = a.W;
  __shared__ uint32_t s_off[kMaxPrepChunks];
  __shared__ uint32_t s_E;
  __shared__ int s_over;
  if (tid == 0) {
    uint64_t run = 0;
    int over = 0;
    for (int i = 0; i < n_chunks; ++i) {
      s_off[i] = (uint32_t)run;
      run += a.chunk_total[(size_t)k * n_chunks + i];
      if (run > 0xFFFFFFFFull) over = 1;
    }
    s_E = (uint32_t)run;
    s_over = over;
  }
  __syncthreads();
  for (int w = tid; w < W; w += kPrepThreads) a.prefix[(size_t)k * (W + 1) + w] += s_off[w / kPrepThreads];
  if (tid == 0) {
    if (s_over) atomicOr(a.status, 1);
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int i = 0; i < n_chunks; ++i) {
      const size_t ci = (size_t)k * n_chunks + i;
      xmin = fmin(xmin, a.chunk_box[4 * ci + 0]); xmax = fmax(xmax, a.chunk_box[4 * ci + 1]);
      ymin = fmin(ymin, a.chunk_box[4 * ci + 2]); ymax = fmax(ymax, a.chunk_box[4 * ci + 3]);
    }
    const double x_ref = a.x_ref[k], y_ref = a.y_ref[k];
    double tr[6];
    trace_coeffs(a.g, x_ref, y_ref, tr);
    const uint32_t E = s_E;
    a.prefix[(size_t)k * (W + 1) + W] = E;
    a.sub[k] = make_sub_info(a, k, x_ref, y_ref, tr, E, xmin, xmax, ymin, ymax);
    count_electrons(a.total_electrons, k, (unsigned long long)E);
  }
}

}  // namespace wayne
