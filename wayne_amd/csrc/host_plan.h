// The HOST side of an exposure's launch plan: everything wayne_exposure_upload works out on the CPU before a kernel
// runs.  No HIP call and no HIP header -- wayne_hip.hip includes this file for the product, and
// tests/native/plan_harness.cpp compiles the SAME file with g++ -fsanitize=address,undefined and drives it from
// property tests on the CPU (tests/test_host_plan.py): the silent-loss defects of rounds 3 and 4 were both here
// (accumulator boxes taken from the first and last ARRAY element of an unordered wavelength grid, commit 081cbbc;
// spectrum estimates that survived wayne_ctx_set_grism, commit d35fc47), behind a boundary only a GPU box could reach.
//
//   SpectrumEstimate     per-bin factors that depend on (grism, wavelengths, stellar flux) alone, cached on their content
//   estimate_thrown      expected electrons per k_lane / k_narrow chunk -> launch order, batches, what k_throw is sized for
//   accumulator_boxes    per read interval: where the thrower's electrons can land (k_ramp loads accumulators only there)
//   plan_psf_apply       the host half of wayne_psf_apply: count reduction, N = (int)(counts ratio), routing, clip rectangle
//   plan_sky             levels of the master sky, alias-table keys, which reads fit a table (k_ramp's sky draw)
//   build_sky_alias      one Walker / Vose table of Poisson(lam)
//
// Reference: the reach of the thrower bounds pyparallel_menu.c:87-108 as the device modes implement it; the trace is
// grism.py:491-506, 779-803 (trace_coeffs, plan_consts.h); the frame offset exposure_generator.py:630-645.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "plan_consts.h"

namespace wayne {
namespace plan {

// Negative-control switches for tests/native (NEVER defined by wayne_amd/build.py): each brings one historical defect
// back so that the harness can show its properties catch it.
#if defined(__HIPCC__) && (defined(WAYNE_PLAN_NEGCTL_ARRAY_ENDS) || defined(WAYNE_PLAN_NEGCTL_STALE_CACHE) || defined(WAYNE_PLAN_NEGCTL_NAN_INTERP))
#error "WAYNE_PLAN_NEGCTL_* are for the CPU harness only"
#endif

constexpr int RNG_SPLIT = 2;   // WAYNE_RNG_SPLIT (include/wayne_hip.h)

// np.interp's precondition on the sensitivity table (grism.py:116-118: "xp must be increasing"; numpy does not check and
// returns nonsense otherwise): finite, non-decreasing wavelengths and finite values.  wayne_ctx_set_grism refuses a table
// that fails it (WAYNE_E_INVALID) -- a bisection over unordered or NaN abscissae has no bracket to find.
inline bool sens_table_ok(const double* swl, const double* sval, int n) {
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(swl[i]) || !std::isfinite(sval[i])) return false;
    if (i > 0 && swl[i] < swl[i - 1]) return false;
  }
  return true;
}

struct SpectrumEstimate {
  // the grism the factors were worked out with
  GrismDev g{};
  bool have_grism = false;
  std::vector<double> sens_wl, sens_val;
  // cache key: the spectrum handed in (every exposure of a visit brings the same wavelengths and stellar flux)
  std::vector<double> wl, flux;
  // rate = flux sens dlam 1e4 1e-3 (electrons per ms at scale 1), the wide fraction and sigma_l of the bin
  std::vector<double> rate, ratio, sigl;
  double smax = 0., wl_lo = 0., wl_hi = 0.;   // largest PSF sigma; smallest and largest wavelength, WHEREVER they sit
  bool sig_ok = false;                        // every sigma and wavelength is a number a bound can be built on
  long rebuilds = 0;                          // (for tests: how often the factors were recomputed)

  void set_grism(const GrismDev& g_, const double* swl, const double* sval, int n_sens) {
    g = g_;
    sens_wl.assign(swl, swl + std::max(n_sens, 0));
    sens_val.assign(sval, sval + std::max(n_sens, 0));
    g.sens_wl = nullptr;   // (device pointers mean nothing here)
    g.sens_val = nullptr;
    have_grism = true;
#ifndef WAYNE_PLAN_NEGCTL_STALE_CACHE
    // what is kept per spectrum was worked out with the previous grism's polynomials and sensitivity
    wl.clear();
    flux.clear();
#endif
  }

  static double poly3(const double* p_, double x) { return ((p_[0] * x + p_[1]) * x + p_[2]) * x + p_[3]; }

  void update(int W, const double* wl_um, const double* fl) {
    if ((int)wl.size() == W && W > 0 && std::memcmp(wl.data(), wl_um, (size_t)W * 8) == 0 &&
        std::memcmp(flux.data(), fl, (size_t)W * 8) == 0)
      return;
    rebuilds += 1;
    wl.assign(wl_um, wl_um + W);
    flux.assign(fl, fl + W);
    rate.assign((size_t)W, 0.); ratio.assign((size_t)W, 0.); sigl.assign((size_t)W, 0.);
    smax = 0.; sig_ok = true; wl_lo = wl_hi = 0.;
    for (int i = 0; i < W; ++i) {
      const double x = wl_um[i];
      double sens = 1.0;
      if (!sens_wl.empty()) {       // np.interp (grism.py:116-118): clamp outside the table, linear inside
#ifdef WAYNE_PLAN_NEGCTL_NAN_INTERP
        if (x <= sens_wl.front()) sens = sens_val.front();     // rounds 1-4: a NaN wavelength passes neither clamp ...
#else
        if (!(x > sens_wl.front())) sens = sens_val.front();   // (a NaN wavelength takes the first value: sig_ok says the rest)
#endif
        else if (x >= sens_wl.back()) sens = sens_val.back();
        else {                                                  // ... and upper_bound(NaN) is end(): sens_val[n], one past the table
          size_t hi = (size_t)(std::upper_bound(sens_wl.begin(), sens_wl.end(), x) - sens_wl.begin());
#ifndef WAYNE_PLAN_NEGCTL_NAN_INTERP
          // (front < x < back, so 1 <= hi <= n - 1 on any table sens_table_ok() passes; a table that does not -- NaN at its
          // end, say -- can send the bisection to either end: the bracket stays inside the table whatever it holds)
          hi = std::min(std::max(hi, (size_t)1), sens_wl.size() - 1);
#endif
          const size_t lo = hi - 1;
          sens = sens_val[lo] + (sens_val[hi] - sens_val[lo]) * (x - sens_wl[lo]) / (sens_wl[hi] - sens_wl[lo]);
        }
      }
      // tools.bin_centers_to_widths (tools.py:106-128): half-gaps to the neighbours, end bins mirror theirs
      double left = 0., right = 0.;
      if (W >= 2) {
        left = (i == 0) ? (wl_um[1] - wl_um[0]) / 2. : (x - wl_um[i - 1]) / 2.;
        right = (i == W - 1) ? (wl_um[W - 1] - wl_um[W - 2]) / 2. : (wl_um[i + 1] - x) / 2.;
      }
      rate[i] = fl[i] * sens * (left + right) * 1e4 * 1e-3;
      ratio[i] = poly3(g.p_ratio, x);
      const double sl = poly3(g.p_sigl, x), sh = poly3(g.p_sigh, x);
      sigl[i] = sl;
      if (!(sl >= 0. && sl < 1e3 && sh >= 0. && sh < 1e3) || !(std::fabs(x) < 1e6)) sig_ok = false;
      if (sl > smax) smax = sl;      // (NaN never passes a comparison: sig_ok already says so)
      if (sh > smax) smax = sh;
#ifdef WAYNE_PLAN_NEGCTL_ARRAY_ENDS
      if (i == 0) wl_lo = x;         // rounds 1-3: "the first and the last wavelength" of the array
      if (i == W - 1) wl_hi = x;
#else
      if (i == 0 || x < wl_lo) wl_lo = x;
      if (i == 0 || x > wl_hi) wl_hi = x;
#endif
    }
  }
};

struct ThrowPlan {
  double est_thrown = 0.;            // electrons k_throw is expected to share out in the longest sub-sample
  double max_chunk_electrons = 0.;   // ... of the fullest k_lane chunk
  double max_narrow = 0.;            // most narrow electrons expected in a bin
  int n_chunks = 0, n_lane_chunks = 0;
  unsigned char chunk_order[kMaxChunks] = {0};   // chunks of kNarrowThreads bins, most electrons first
  unsigned char lane_order[kMaxChunks] = {0};    // chunks of kLaneThreads bins, most electrons first
};

// Expected number of electrons k_throw shares out (split mode: only the bins beyond a lane's cap, normally none) in the
// longest sub-sample of an exposure (the counts chain of k_prep_wl / k_prep_sub without its Poisson noise and transit
// depth): sizes the thrower's grid, nothing else -- the kernel distributes the electrons it actually finds.
inline void estimate_thrown(SpectrumEstimate& e, int W, const double* wl_um, const double* flux, int K,
                            const double* dur_ms, double scale_factor, int rng_mode, ThrowPlan* out) {
  const int n_chunks = (W + kNarrowThreads - 1) / kNarrowThreads;
  const int n_lane_chunks = (W + kLaneThreads - 1) / kLaneThreads;
  std::vector<double> chunk_e((size_t)n_chunks, 0.), lane_e((size_t)n_lane_chunks, 0.);
  double dur_max = 0.;
  for (int k = 0; k < K; ++k) dur_max = std::max(dur_max, dur_ms[k]);
  e.update(W, wl_um, flux);
  const double per_ms = dur_max * scale_factor;
  double total = 0.;
  out->max_narrow = 0.;
  for (int i = 0; i < W; ++i) {
    double cnt = e.rate[i] * per_ms;
    if (!(cnt > 0.)) continue;
    chunk_e[(size_t)(i / kNarrowThreads)] += cnt;
    lane_e[(size_t)(i / kLaneThreads)] += cnt;
    if (rng_mode == RNG_SPLIT) {
      const double wide = std::floor(std::min(std::max(cnt * e.ratio[i], 0.), cnt));
      out->max_narrow = std::max(out->max_narrow, cnt - wide);
      const double sl = e.sigl[i];
      if (cnt - wide >= (double)kSplitMin && cnt - wide <= (double)kSplitMaxNarrow && sl > 0.05 &&
          sl * 6.5 <= (double)kNarrowR) cnt = wide;   // narrow part: k_narrow
      if (cnt <= 0.9 * (double)kLaneMax) cnt = 0.;    // thrown by the bin's own lane (k_lane); 10 % headroom for the noise
    }
    total += cnt;
  }
  // chunks by expected electrons, most first (stable for ties; NaN sums -- hostile input -- sort as "no electrons")
  auto order_of = [](const std::vector<double>& v, unsigned char* dst) {
    std::vector<int> order(v.size());
    for (size_t i = 0; i < v.size(); ++i) order[i] = (int)i;
    auto key = [&](int i) { return v[(size_t)i] == v[(size_t)i] ? v[(size_t)i] : -1.; };
    std::stable_sort(order.begin(), order.end(), [&](int a_, int b_) { return key(a_) > key(b_); });
    for (size_t i = 0; i < v.size() && i < (size_t)kMaxChunks; ++i) dst[i] = (unsigned char)order[i];
  };
  order_of(chunk_e, out->chunk_order);
  order_of(lane_e, out->lane_order);
  out->max_chunk_electrons = 0.;
  for (double x : lane_e) out->max_chunk_electrons = std::max(out->max_chunk_electrons, x);
  out->n_chunks = n_chunks;
  out->n_lane_chunks = n_lane_chunks;
  out->est_thrown = total;
}

// k_lane's batches: enough workgroups to fill the chip several times over (~2048), no more -- a finely sampled scan
// (K in the thousands) otherwise launches tens of thousands of workgroups of ~1000 electrons each.  thin: the expected
// electrons of the fullest chunk in the longest sub-sample fit the flush list with room to spare.
inline void lane_batches(int K, int W, double max_chunk_electrons, int* kb, bool* thin) {
  const int n_chunks_l = (W + kLaneThreads - 1) / kLaneThreads;
  int b = (int)(((long long)K * n_chunks_l) / 2048);
  *kb = std::min(std::max(b, 1), kLaneBatchMax);
  *thin = max_chunk_electrons <= 0.9 * kLaneListCap;
}

// Where can the accumulators of read interval r be non-zero after the thrower?  An electron lands within
// sigma sqrt(2 ln 2^34) = 6.87 sigma of its bin in every rng mode (k_lane: "a tile that holds every electron"; k_throw's
// per-electron mode reaches 6.76 sigma, the replay thrower's rand_r / RAND_MAX 6.56 sigma; k_narrow's window is +-6 px),
// and the bins of a sub-sample lie on the straight trace between its smallest and its largest wavelength.  So per read:
// the union over its sub-samples of the trace's end points, +- (6.9 sigma_max + 2) px, in bordered coordinates.  Returns
// false (-> k_ramp loads everything) when the numbers are not ones a bound can be built on.
inline bool accumulator_boxes(SpectrumEstimate& e, int W, const double* wl_um, const double* flux, int K, int R, int S,
                              int sub_scale, const double* x_ref, const double* y_ref, const int32_t* sample_read,
                              int (*box)[4]) {
  const GrismDev& g = e.g;
  e.update(W, wl_um, flux);      // (the ABI does not require increasing wavelengths: smallest and largest present, wherever they sit)
  if (!e.sig_ok) return false;
  // + 1 px: the ends of the trace are taken at the four corners of the rectangle a read's star positions span, not at
  // every sub-sample (thousands on a finely sampled scan).  The end points move monotonically with the star (d end / d
  // star = 1 + O(1e-3)); what a corner can miss is the curvature of the trace polynomials over the rectangle -- their
  // second derivatives are ~1e-8 / px^2, a scan is a few hundred pixels long: < 0.01 px
  const double reach = 6.9 * e.smax + 2. + 1.;
  if (!(reach < 400.)) return false;
  double lo_x[16], hi_x[16], lo_y[16], hi_y[16];
  bool any[16];
  for (int r = 0; r < 16; ++r) { any[r] = false; lo_x[r] = lo_y[r] = 0.; hi_x[r] = hi_y[r] = 0.; }
  for (int k = 0; k < K; ++k) {
    const int r = sample_read[k];
    if (r < 0 || r >= R || r >= 16) return false;
    const double xr = x_ref[k], yr = y_ref[k];
    if (!(std::fabs(xr) < 1e6 && std::fabs(yr) < 1e6)) return false;
    if (!any[r]) { any[r] = true; lo_x[r] = hi_x[r] = xr; lo_y[r] = hi_y[r] = yr; }
    else {
      lo_x[r] = std::min(lo_x[r], xr); hi_x[r] = std::max(hi_x[r], xr);
      lo_y[r] = std::min(lo_y[r], yr); hi_y[r] = std::max(hi_y[r], yr);
    }
  }
  for (int r = 0; r < 16; ++r) {
    box[r][0] = box[r][2] = 0x3FFFFFFF; box[r][1] = box[r][3] = -0x3FFFFFFF;
    if (!any[r]) { box[r][0] = box[r][1] = box[r][2] = box[r][3] = 0; continue; }   // a read without sub-samples
    for (int corner = 0; corner < 4; ++corner) {
      const double xr = (corner & 1) ? hi_x[r] : lo_x[r], yr = (corner & 2) ? hi_y[r] : lo_y[r];
      double tr[6];
      trace_coeffs(g, xr, yr, tr);
      for (int end = 0; end < 2; ++end) {
        const double wl = end ? e.wl_hi : e.wl_lo;
        const double x = (wl - tr[5]) / tr[4];
        const double y = tr[0] * (x - xr) + tr[1] + yr;
        const double xs = x - (double)sub_scale + kBorder, ys = y - (double)sub_scale + kBorder;
        if (!(std::fabs(xs) < 1e6 && std::fabs(ys) < 1e6)) return false;
        box[r][0] = std::min(box[r][0], (int)std::floor(xs - reach));
        box[r][1] = std::max(box[r][1], (int)std::floor(xs + reach) + 1);
        box[r][2] = std::min(box[r][2], (int)std::floor(ys - reach));
        box[r][3] = std::max(box[r][3], (int)std::floor(ys + reach) + 1);
      }
    }
    box[r][0] = std::max(box[r][0], 0); box[r][2] = std::max(box[r][2], 0);
    box[r][1] = std::min(box[r][1], S); box[r][3] = std::min(box[r][3], S);
  }
  return true;
}

// The host half of wayne_psf_apply (the inner drop-in boundary: pyparallel.pyx:14-38 -> pyparallel_menu.c:10-113): the
// count reduction A1 with the reference's silent int overflows turned into errors, N = (int)(counts * ratio) per bin
// (:89, with the C cast's behaviour at NaN and beyond int spelled out), the split mode's routing of every bin -- the rule
// of k_prep_sub -- and the clip rectangle of the throwers' tiles.  Every electron of the input must be routed exactly
// once: nsplit + nlane + (prefix[i + 1] - prefix[i]) == counts[i].
struct PsfPlan {
  std::vector<uint32_t> prefix;                  // [size + 1] exclusive prefix of the electrons k_throw shares out
  std::vector<int32_t> nwide, nsplit, nlane;     // [size]
  bool any_split = false, any_lane = false;
  uint32_t run = 0;                              // electrons for k_throw
  long long total = 0;                           // sum(counts)
  int tx0 = 0, ty0 = 0, tw = 0, th = 0;          // clip region of the tiles (frame coordinates); tw = 0: none
};
enum PsfPlanError { PSF_OK = 0, PSF_NEGATIVE = 1, PSF_OVERFLOW_REPLAY = 2, PSF_OVERFLOW_TOTAL = 3 };

inline int plan_psf_apply(const int32_t* counts, int size, const double* x_pos, const double* y_pos, const double* psf_ratio,
                          const double* psf_sigmal, int N, int rng_mode, int threads_compat, int margin, PsfPlan* out) {
  *out = PsfPlan();
  long long total = 0;
  for (int i = 0; i < size; ++i) {
    if (counts[i] < 0) return PSF_NEGATIVE;
    total += counts[i];
  }
  out->total = total;
  if (rng_mode == 0 && total * (long long)threads_compat > 2147483647LL) return PSF_OVERFLOW_REPLAY;   // (pyparallel_menu.c:12,48)
  if (total > 0xFFFFFFFFLL) return PSF_OVERFLOW_TOTAL;
  out->prefix.assign((size_t)size + 1, 0u);
  out->nwide.assign((size_t)size, 0); out->nsplit.assign((size_t)size, 0); out->nlane.assign((size_t)size, 0);
  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  uint32_t run = 0;
  for (int i = 0; i < size; ++i) {
    out->prefix[(size_t)i] = run;
    const double nw = (double)counts[i] * psf_ratio[i];  // N = counts*ratio (:89)
    out->nwide[(size_t)i] = (nw >= 2147483647.) ? 2147483647
                            : (nw <= -2147483648.) ? (int32_t)(-2147483647 - 1)
                            : (!(nw == nw))        ? (int32_t)(-2147483647 - 1)
                                                   : (int32_t)nw;
    uint32_t thrown = (uint32_t)counts[i];
    if (rng_mode == RNG_SPLIT) {      // same rule as k_prep_sub
      const uint32_t wide = (uint32_t)std::min<int64_t>(std::max(out->nwide[(size_t)i], 0), counts[i]);
      const uint32_t narrow = (uint32_t)counts[i] - wide;
      const bool split = narrow >= (uint32_t)kSplitMin && narrow <= kSplitMaxNarrow && psf_sigmal[i] > 0.05 &&
                         psf_sigmal[i] * 6.5 <= (double)kNarrowR;
      const uint32_t ind = split ? wide : (uint32_t)counts[i];
      const bool lane = ind <= (uint32_t)kLaneMax;
      out->nsplit[(size_t)i] = split ? (int32_t)narrow : 0;
      out->nlane[(size_t)i] = lane ? (int32_t)ind : 0;
      thrown = lane ? 0u : ind;
      if (out->nsplit[(size_t)i] > 0) out->any_split = true;
      if (out->nlane[(size_t)i] > 0) out->any_lane = true;
    }
    run += thrown;
    if (counts[i] > 0 && std::isfinite(x_pos[i]) && std::isfinite(y_pos[i])) {
      xmin = std::min(xmin, x_pos[i]); xmax = std::max(xmax, x_pos[i]);
      ymin = std::min(ymin, y_pos[i]); ymax = std::max(ymax, y_pos[i]);
    }
  }
  out->prefix[(size_t)size] = run;
  out->run = run;
  if (xmax >= xmin) {
    auto clampi = [](double v) { return (int)std::min(std::max(v, -1e6), 1e6); };
    const int x0 = std::max(clampi(std::floor(xmin)) - margin, 1), x1 = std::min(clampi(std::floor(xmax)) + margin + 1, N);
    const int y0 = std::max(clampi(std::floor(ymin)) - margin, 1), y1 = std::min(clampi(std::floor(ymax)) + margin + 1, N);
    if (x1 > x0 && y1 > y0) { out->tx0 = x0; out->ty0 = y0; out->tw = x1 - x0; out->th = y1 - y0; }   // clip region of the tiles
  }
  return PSF_OK;
}

// Does Poisson(lam) fit an alias table of kSkyAlias entries (mass beyond the table < 1e-14)?
inline bool sky_alias_fits(double lam) {
  return lam >= 0. && lam + 8. * std::sqrt(lam) + 8. <= (double)(kSkyAlias - 1);
}

// Walker / Vose alias table of Poisson(lam) over 0 .. kSkyAlias-1, entry = alias << 24 | threshold:
// a 32-bit word w selects column w >> 24 and keeps it when (w & 0xFFFFFF) < threshold, else takes the
// alias.  Probabilities in fp64, thresholds rounded to 24 bits (the resolution of a float32 uniform).
inline void build_sky_alias(double lam, uint32_t* out /* kSkyAlias */) {
  constexpr int n = kSkyAlias;
  double q[n], prob[n];
  int alias[n], small[n], large[n], n_small = 0, n_large = 0;
  double sum = 0.;
  for (int k = 0; k < n; ++k) q[k] = 0.;
  if (!(lam > 0.)) { q[0] = 1.; sum = 1.; }
  else {
    // pmf by recurrence from the mode (one exp / log / lgamma per table): p(k+1) = p(k) lam / (k+1)
    const int k0 = std::min((int)lam, n - 1);
    q[k0] = std::exp(-lam + k0 * std::log(lam) - std::lgamma(k0 + 1.0));
    for (int k = k0; k + 1 < n; ++k) q[k + 1] = q[k] * lam / (double)(k + 1);
    for (int k = k0; k > 0; --k) q[k - 1] = q[k] * (double)k / lam;
    for (int k = 0; k < n; ++k) sum += q[k];
  }
  for (int k = 0; k < n; ++k) {
    q[k] = q[k] / sum * n;
    if (q[k] < 1.) small[n_small++] = k; else large[n_large++] = k;
    prob[k] = 1.;
    alias[k] = k;
  }
  while (n_small > 0 && n_large > 0) {
    const int s_ = small[--n_small];
    const int l_ = large[--n_large];
    prob[s_] = q[s_];
    alias[s_] = l_;
    q[l_] = (q[l_] + q[s_]) - 1.;
    if (q[l_] < 1.) small[n_small++] = l_; else large[n_large++] = l_;
  }
  for (int k = 0; k < n; ++k) {
    double t = std::floor(prob[k] * 16777216. + 0.5);
    if (t > 16777215.) t = 16777215.;
    if (t < 0.) t = 0.;
    out[k] = ((uint32_t)alias[k] << 24) | (uint32_t)t;
  }
}

// The sky draws of an exposure (k_ramp, sky_draw): levels of the master sky, one alias table of
// Poisson(level * bg_count) per level and distinct read interval.
struct SkyPlan {
  bool alias_on = false;     // every read fits its tables (otherwise the exposure takes the direct sampler)
  bool pieces = false;       // some pixel's remainder can exceed kSkyPiece: drawn in pieces
  uint32_t mask = 0;         // bit r: read r's rates fit a table
  int L = 1;                 // levels
  float level[16] = {0};     // ascending; [0] = the smallest positive sky pixel
  unsigned char tab0[16] = {0};   // first table of read r
  std::vector<uint32_t> keys;     // bit pattern of the float32 rate of table t = (distinct interval j) * L + level l
  std::vector<char> fits;         // per distinct interval
  int n_bg = 0;
};

inline void plan_sky(double sky_ct_s, int R, const double* read_dt_s, bool has_sky, float sky_min, float sky_max,
                     const std::vector<float>& sky_sorted, SkyPlan* p) {
  *p = SkyPlan();
  for (float& l_ : p->level) l_ = sky_min;
  if (!(sky_ct_s > 0. && has_sky && sky_max > 0.f) || sky_sorted.empty() || R < 1 || R > kMaxReads) return;
  // distinct read intervals (float32 bg_count, as the kernel and numpy use it, exposure_generator.py:489-493)
  std::vector<float> bg;            // distinct bg_count values, first-appearance order
  std::vector<int> bg_of((size_t)R);
  for (int r = 0; r < R; ++r) {
    const float b = (float)(sky_ct_s * read_dt_s[r]);
    size_t j = 0;
    while (j < bg.size() && std::memcmp(&bg[j], &b, 4) != 0) ++j;
    if (j == bg.size()) bg.push_back(b);
    bg_of[r] = (int)j;
  }
  p->n_bg = (int)bg.size();
  const int L = std::max(1, std::min(kMaxReads / (int)bg.size(), kMaxReads));
  // levels = the l/L quantiles of the positive sky pixels (actual pixel values, [0] = the minimum): most
  // pixels sit just above their level, so their own remainder is a fraction of an electron
  float levels[16];
  for (int l = 0; l < 16; ++l) levels[l] = sky_max;
  for (int l = 0; l < L; ++l) levels[l] = sky_sorted[(size_t)l * sky_sorted.size() / (size_t)L];
  p->keys.assign(bg.size() * (size_t)L, 0u);
  p->fits.assign(bg.size(), 1);
  for (size_t j = 0; j < bg.size(); ++j)
    for (int l = 0; l < L; ++l) {
      const float lam = levels[l] * bg[j];
      if (!sky_alias_fits((double)lam)) p->fits[j] = 0;
      std::memcpy(&p->keys[j * L + l], &lam, 4);
    }
  uint32_t mask = 0;
  for (int r = 0; r < R; ++r) {
    if (p->fits[(size_t)bg_of[r]]) mask |= 1u << r;
    p->tab0[r] = (unsigned char)(bg_of[r] * L);
  }
  p->mask = mask;
  if (mask != (1u << R) - 1u) return;
  p->alias_on = true;
  p->L = L;
  for (int l = 0; l < 16; ++l) p->level[l] = levels[l];
  // largest remainder any pixel can have: the widest gap between levels (the top one reaches sky_max)
  float gap = sky_max - levels[L - 1];
  for (int l = 0; l + 1 < L; ++l) gap = std::max(gap, levels[l + 1] - levels[l]);
  float bg_max = 0.f;
  for (float b : bg) bg_max = std::max(bg_max, b);
  p->pieces = !(gap * bg_max <= kSkyPiece);
}

}  // namespace plan
}  // namespace wayne
