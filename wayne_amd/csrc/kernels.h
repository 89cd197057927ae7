// HIP kernels (gfx950) of the WFC3-IR exposure-synthesis path.
//
//   k_prep_wl      per-wavelength arrays: PSF polynomials, sensitivity LUT, bin widths   (A8, A9)
//   k_prep_sub     per sub-sample: trace, bin positions, expected counts, Poisson/round,
//                  sigma split, exclusive prefix of counts, LDS tile rectangle            (A6, A7, A9, A10)
//   k_throw        the electron thrower: LDS int32 tile per (sub-sample, split),
//                  flushed x flat into the read-interval accumulator                     (A1-A4, A11, A12)
//   k_cosmic       cosmic-ray hits per read interval                                      (A13, cosmic_rays.py)
//   k_ramp         fused up-the-ramp kernel: sky, gain, cumulative, dark, non-linearity,
//                  clip, reference pixels, zero read, read noise                          (A13-A15)
//
// "A<n>" are the row ids of SURVEY.md section 8(a); reference file:line
// citations are next to each formula.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "philox.h"
#include "samplers.h"

namespace wayne {

constexpr int kBorder = 5;            // reference-pixel border (detector.py:146-147)
constexpr int kQBits = 28;            // accumulator fixed point: 2^28 per electron
constexpr double kQ = 268435456.0;    //   (int64: 3.4e10 e- of range per pixel and read interval,
constexpr double kInvQ = 1.0 / 268435456.0;  // 1.9e-9 e- rounding per tile flush)
constexpr double kGain = 2.35;        // detector.py:30
constexpr double kReadNoise = 14.1 / 2.35;  // detector.py:33
constexpr double kMinCounts = -20.0;  // detector.py:26
constexpr double kMaxCounts = 78000.0;  // detector.py:28
constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------
// shared device structs
// ---------------------------------------------------------------------------
struct GrismDev {
  double trace[9], wlsol[9];
  double p_ratio[4], p_sigl[4], p_sigh[4];
  double flat_wmin, flat_wmax;
  int n_sens;
  const double* sens_wl;
  const double* sens_val;
};

// Per sub-sample record written by k_prep_sub and read by k_throw.
struct SubInfo {
  uint32_t electrons;      // E_k
  int tx0, ty0, tw, th;    // LDS tile rectangle, frame coordinates
  int read;                // read interval this sub-sample accumulates into
  int replay_seed;         // the reference's `test`
  int pad_;
  double x_ref, y_ref;     // star position of the sub-sample (full-frame coords)
  double a_t_i, a_w, b_w;  // 1/m_t, m_w, c_w for the flat (grism.py:365-372)
  double inv_norm;         // 1 / sqrt(a_t_i^2 + 1)
};

struct WlArrays {   // all [W]
  double* ratio;    // psf_ratio_poly(wl)   (fp64: the sigma split is done in fp64)
  double* sigl;     // psf_sigmal_poly(wl)
  double* sigh;     // psf_sigmah_poly(wl)
  double* sens;     // np.interp(wl, throughput_wl, throughput_val)
  double* dlam;     // tools.bin_centers_to_widths(wl)
};

__device__ __forceinline__ double poly3(const double* c, double x) {
  // np.poly1d([c0,c1,c2,c3])(x): Horner, highest power first (grism.py:85-90,113-115)
  return ((c[0] * x + c[1]) * x + c[2]) * x + c[3];
}

// ---------------------------------------------------------------------------
// k_prep_wl : A8 + the wavelength-only part of A9
// ---------------------------------------------------------------------------
__global__ void k_prep_wl(GrismDev g, int W, const double* __restrict__ wl, WlArrays o, uint32_t* misc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the exposure's status words (total electrons, overflow flag) start from zero: cleared here, by the
  // first kernel of the exposure, instead of by a separate fill in front of it
  if (i < 16) misc[i] = 0u;
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
  } else if (x <= g.sens_wl[0]) {
    s = g.sens_val[0];
  } else if (x >= g.sens_wl[n - 1]) {
    s = g.sens_val[n - 1];
  } else {
    int lo = 0, hi = n - 1;  // sens_wl[lo] <= x < sens_wl[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (g.sens_wl[mid] <= x) lo = mid; else hi = mid;
    }
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
// k_lightcurve : transit-depth matrix depth[K][W] on the device
// ---------------------------------------------------------------------------
// Replaces Observation.generate_lightcurves (observation.py:293-357: one
// pylightcurve.transit + one pylightcurve.eclipse call per wavelength element
// per exposure).  Star with Claret limb darkening I(mu) = 1 - sum a_n (1 - mu^(n/2))
// occulted by a disk of radius p at separation z:
//   dF = int_0^{p-z} I 2 pi r dr  +  int_{|z-p|}^{min(1,z+p)} I(r) r theta(r) dr,
//   theta = 4 atan2(sqrt(p^2 - (r-z)^2), sqrt((r+z)^2 - p^2))
// with a 24-node tanh-sinh rule (wayne_amd/lightcurve.py states the same model
// in numpy).  float32 integrand in cancellation-free form, float64 sum.
constexpr int kLcNodes = 24;
struct LcArgs {
  int K, W;
  const double* z;        // [K]
  const double* hidden;   // [K] or null
  const double* rp;       // [W]
  double ld[4];
  double f0;              // pi (1 - sum a_n n/(n+4))
  float x[kLcNodes], w[kLcNodes], d[kLcNodes];   // node, weight, distance to the nearer end
  double p_lo, p_hi;      // range of rp over the W wavelengths
  double* depth;          // [K*W]
};

__device__ __forceinline__ double lc_prim(const double* a, double m) {
  // int I(m) m dm = m^2/2 (1 - sum a_n) + a1 m^2.5/2.5 + a2 m^3/3 + a3 m^3.5/3.5 + a4 m^4/4
  const double s = sqrt(m);
  const double m2 = m * m;
  return (m2 / 2.) * (1. - a[0] - a[1] - a[2] - a[3]) + a[0] * m2 * s / 2.5 + a[1] * m2 * m / 3. +
         a[2] * m2 * m * s / 3.5 + a[3] * m2 * m2 / 4.;
}

// 1 - transit for one (z, p): the quadrature described above.
__device__ __forceinline__ double lc_deficit(const LcArgs& a, double z, double p) {
  if (!(z < 1. + p)) return 0.;
  const double r_full = fmin(fmax(p - z, 0.), 1.);
  const double mu_f = sqrt(1. - r_full * r_full);
  double dF = 2. * kPi * (lc_prim(a.ld, 1.) - lc_prim(a.ld, mu_f));
  const double ra = fabs(z - p), rb = fmin(1., z + p);
  if (rb > ra) {
    const float L = (float)(rb - ra), raf = (float)ra, zf = (float)z, pf = (float)p;
    const float gap = (float)(1. - rb);            // 1 - rb >= 0
    const float a1 = (float)a.ld[0], a2 = (float)a.ld[1], a3 = (float)a.ld[2], a4 = (float)a.ld[3];
    double sum = 0.;
#pragma unroll 4
    for (int i = 0; i < kLcNodes; ++i) {
      const float x = a.x[i], dn = a.d[i];
      const float lo = L * (x < 0.5f ? dn : 1.f - dn);    // r - ra
      const float hi = L * (x < 0.5f ? 1.f - dn : dn);    // rb - r
      const float r = raf + lo;
      // p^2 - (r - z)^2 and (r + z)^2 - p^2 without cancellation:
      //   |z - p| = ra  =>  p^2 - (r-z)^2 = (p - |r - z|)(p + |r - z|), and p - |r-z| vanishes at r = ra only
      const float rmz = r - zf;
      const float num = fmaxf((pf - fabsf(rmz)) * (pf + fabsf(rmz)), 0.f);
      const float den = fmaxf((r + zf - pf) * (r + zf + pf), 0.f);
      const float theta = 4.f * atan2f(sqrtf(num), sqrtf(den));
      const float mu = sqrtf(fmaxf((gap + hi) * (1.f + r), 0.f));   // sqrt((1-r)(1+r))
      const float sm = sqrtf(mu);
      const float I = 1.f - a1 * (1.f - sm) - a2 * (1.f - mu) - a3 * (1.f - mu * sm) - a4 * (1.f - mu * mu);
      sum += (double)(I * r * theta * a.w[i]);
    }
    dF += sum * (double)L;
  }
  return dF / a.f0;
}

// One workgroup per sub-sample.  The radius ratios of a spectrum span a narrow
// interval [p_lo, p_hi] and, at fixed z, the deficit is an analytic function of p except where the
// geometry changes regime (p = |1 - z|: a contact; p = z: the planet reaches the centre).  So the
// quadrature is evaluated at the kLcCheb Chebyshev-Lobatto points of the interval only and every
// wavelength evaluates the Chebyshev interpolant (its error is far below the quadrature's 2e-8) -- unless a regime change falls inside the interval for this z, in which case
// every wavelength is integrated on its own as before.
constexpr int kLcCheb = 16;   // points: t_j = cos(j pi / (kLcCheb - 1))

__global__ __launch_bounds__(256) void k_lightcurve(LcArgs a) {
  static_assert(kLcCheb * kLcCheb == 256, "one thread per entry of the cosine table");
  constexpr int n = kLcCheb - 1;
  const int k = blockIdx.x;            // one workgroup per sub-sample: the node values are computed once
  __shared__ double s_f[kLcCheb];      // samples at the Lobatto points
  __shared__ double s_c[kLcCheb];      // Chebyshev coefficients (first and last halved)
  __shared__ double s_cos[kLcCheb][kLcCheb];
  const double z = a.z[k];
  const double p_lo = a.p_lo, p_hi = a.p_hi;
  const double width = p_hi - p_lo, guard = 1e-9 + 1e-6 * width;
  auto inside = [&](double v) { return v > p_lo - guard && v < p_hi + guard; };
  const bool no_transit = !(z < 1. + p_lo) && !(z < 1. + p_hi);
  const bool direct = !no_transit && (inside(fabs(1. - z)) || inside(z) || inside(z - 1.));
  const bool flat = width <= 1e-14 * p_hi;
  if (!direct && !no_transit) {
    // samples -> Chebyshev coefficients by the discrete cosine sum (end terms halved); every wavelength
    // then evaluates the series with Clenshaw's recurrence: 16 multiply-adds, no divisions
    const int tm = threadIdx.x / kLcCheb, tj = threadIdx.x % kLcCheb;
    s_cos[tm][tj] = cos((double)(tm * tj) * kPi / (double)n);
    if (threadIdx.x < kLcCheb) {
      const double t = cos((double)threadIdx.x * kPi / (double)n);
      s_f[threadIdx.x] = lc_deficit(a, z, 0.5 * (p_lo + p_hi) + 0.5 * width * t);
    }
    __syncthreads();
    if (threadIdx.x < kLcCheb) {
      const int m = threadIdx.x;
      double acc = 0.5 * (s_f[0] * s_cos[m][0] + s_f[n] * s_cos[m][n]);
      for (int j = 1; j < n; ++j) acc += s_f[j] * s_cos[m][j];
      acc *= 2. / (double)n;
      s_c[m] = (m == 0 || m == n) ? 0.5 * acc : acc;
    }
    __syncthreads();
  }
  const double hid = a.hidden ? a.hidden[k] : 0.;
  for (int w = threadIdx.x; w < a.W; w += blockDim.x) {
    const double p = a.rp[w];
    double deficit = 0.;   // 1 - transit
    if (direct) {
      deficit = lc_deficit(a, z, p);
    } else if (!no_transit) {
      if (flat) {
        deficit = s_f[0];
      } else {
        const double t = (2. * p - (p_lo + p_hi)) / width;
        double b1 = 0., b2 = 0.;
#pragma unroll
        for (int m = n; m >= 1; --m) {
          const double b0 = s_c[m] + 2. * t * b1 - b2;
          b2 = b1;
          b1 = b0;
        }
        deficit = s_c[0] + t * b1 - b2;
        if (!(p == p)) deficit = 0.;   // a NaN radius ratio (negative depth in the input spectrum): no transit, as before
      }
    }
    // eclipse term: (1 - eclipse) = f hidden / (1 + f), f = planet_spectrum = p^2 (observation.py:352-355)
    double ecl = 0.;
    if (a.hidden) {
      const double f = p * p;
      ecl = f * hid / (1. + f);
    }
    a.depth[(size_t)k * a.W + w] = deficit + ecl;
  }
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
  double scale_factor;
  const double* wl;          // [W]
  const double* flux;        // [W]
  const double* depth;       // [K*W] or null
  const double* x_ref;       // [K]
  const double* y_ref;       // [K]
  const double* dur_ms;      // [K]
  const int32_t* replay_seed;  // [K]
  const int32_t* sample_read;  // [K]
  WlArrays wa;
  // outputs
  int32_t* counts;           // [K*W]
  int32_t* nwide;            // [K*W]
  int32_t* nsplit;           // [K*W] split mode: > 0 narrow electrons handed to k_narrow's multinomial,
                             //        < 0 minus the electrons of a sparse bin (all thrown by k_narrow), else 0
  uint32_t* prefix;          // [K*(W+1)] exclusive prefix of the electrons k_throw throws one by one
  double* xpos;              // [K*W] frame coords (x_sub)
  double* ypos;              // [K*W]
  SubInfo* sub;              // [K]
  unsigned long long* total_electrons;  // += E_k
  int* status;               // set non-zero on overflow
  uint32_t* chunk_total;     // [K * n_chunks] electrons (for k_throw) per chunk of kPrepThreads bins
  double* chunk_box;         // [K * n_chunks * 4] xmin, xmax, ymin, ymax of the chunk's populated bins
};

constexpr int kPrepThreads = 512;
constexpr int kNarrowR = 6;            // k_narrow window: +-6 pixels about the bin's pixel (>= 6.5 sigma_l)
constexpr int kSparseMax = 16;         // WAYNE_RNG_SPLIT: bins with fewer electrons are thrown lane-per-bin (k_narrow)

__device__ __forceinline__ void trace_coeffs(const GrismDev& g, double x_ref, double y_ref, double* o) {
  // o = {m_t, c_t, m_w, c_w, m_wl, c_wl}
    // wavelength_calibration_coeffs (grism.py:779-803)
    const double* t = g.trace;
    const double* b = g.wlsol;
    const double m_t = t[3] + t[4] * x_ref + t[5] * y_ref + t[6] * (x_ref * x_ref) +
                       t[7] * x_ref * y_ref + t[8] * (y_ref * y_ref);
    const double c_t = t[0] + t[1] * x_ref + t[2] * y_ref;
    const double m_w = b[3] + b[4] * x_ref + b[5] * y_ref + b[6] * (x_ref * x_ref) +
                       b[7] * x_ref * y_ref + b[8] * (y_ref * y_ref);
    const double c_w = (b[0] + b[1] * x_ref) + b[2] * y_ref;
    // _get_x_to_wl_poly_coeffs (grism.py:553-602): line through the trace
    // points at x_ref+10 and x_ref+20, wavelength in micron.
    const double xa = x_ref + 10, xb = x_ref + 20;
    const double ya = m_t * (xa - x_ref) + c_t + y_ref;  // x_to_y (grism.py:537)
    const double yb = m_t * (xb - x_ref) + c_t + y_ref;
    const double da = sqrt((ya - y_ref) * (ya - y_ref) + (xa - x_ref) * (xa - x_ref));
    const double db = sqrt((yb - y_ref) * (yb - y_ref) + (xb - x_ref) * (xb - x_ref));
    const double wa_ = (m_w * da + c_w) * 1e-4;  // angstrom -> micron
    const double wb_ = (m_w * db + c_w) * 1e-4;
    const double m_wl = (wb_ - wa_) / (xb - xa);
    const double c_wl = wa_ - m_wl * xa;
    o[0] = m_t; o[1] = c_t; o[2] = m_w; o[3] = c_w; o[4] = m_wl; o[5] = c_wl;
}

// One workgroup per (sub-sample, chunk of kPrepThreads bins): positions, counts,
// sigma split, and the chunk-local exclusive prefix; k_prep_fix then adds the
// chunk offsets.  K * ceil(W / 512) workgroups instead of K: the whole chip works.
__global__ __launch_bounds__(kPrepThreads) void k_prep_sub(PrepArgs a) {
  const int k = blockIdx.x;
  const int ch = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int W = a.W;
  constexpr int NW = kPrepThreads / 64;
  __shared__ double s_tr[8];        // m_t, c_t, m_w, c_w, m_wl, c_wl
  __shared__ uint32_t s_wsum[NW];   // per-wave totals
  __shared__ double s_red[4][NW];

  const double x_ref = a.x_ref[k], y_ref = a.y_ref[k];
  if (tid == 0) trace_coeffs(a.g, x_ref, y_ref, s_tr);
  __syncthreads();
  const double m_t = s_tr[0], c_t = s_tr[1], m_wl = s_tr[4], c_wl = s_tr[5];
  const double dur = a.dur_ms[k];
  const bool noisy = (a.flags & (1u << 5)) != 0;  // WAYNE_F_ADD_STELLAR_NOISE

  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  bool overflow = false;
  unsigned long long n_split_total = 0;   // per thread

  const int w = ch * kPrepThreads + tid;
  uint32_t c = 0;
  if (w < W) {
      const double wl = a.wl[w];
    // wl_to_x / wl_to_y (grism.py:651, 667-669), then the sub-array shift
    // x_sub = x_pos - sub_scale (exposure_generator.py:630-632)
    const double x = (wl - c_wl) / m_wl;
    const double y = m_t * (x - x_ref) + c_t + y_ref;
    const double xs = x - (double)a.sub_scale;
    const double ys = y - (double)a.sub_scale;
    a.xpos[(size_t)k * W + w] = xs;
    a.ypos[(size_t)k * W + w] = ys;
    // counts chain (exposure_generator.py:344-348, 602-628, 649-687):
    //   F (1 - depth) * Sens * dlam[um] * 1e4 [A/um] * dur[ms] * 1e-3 [s/ms] * scale
    double f = a.flux[w];
    if (a.depth) f = f * (1. - a.depth[(size_t)k * W + w]);
    double lam = f * a.wa.sens[w];
    lam = lam * a.wa.dlam[w];
    lam = lam * 1e4;
    lam = lam * dur;
    lam = lam * 1e-3;
    lam = lam * a.scale_factor;
    double cnt;
    if (noisy) {
      PhiloxStream rng(a.seed, STAGE_COUNTS, (uint32_t)w, (uint32_t)k, a.exposure);
      cnt = poisson<ExactMath<double> >(lam, rng);   // np.random.poisson (:626)
    } else {
      cnt = rint(lam);                      // np.round, half to even (:628)
    }
    if (!(cnt >= 0.)) cnt = 0.;             // negative / NaN flux throws no electrons
    if (cnt > 2147483647.) { cnt = 2147483647.; overflow = true; }
    c = (uint32_t)cnt;
    a.counts[(size_t)k * W + w] = (int32_t)c;
    // N = counts*psf_ratio truncated (pyparallel_menu.c:89), in fp64
    double nw = (double)(int32_t)c * a.wa.ratio[w];
    int32_t nwi = (nw >= 2147483647.) ? 2147483647 : (nw <= -2147483648.) ? (int32_t)(-2147483647 - 1) : (int32_t)nw;
    a.nwide[(size_t)k * W + w] = nwi;
    if (c > 0) {
      xmin = fmin(xmin, xs); xmax = fmax(xmax, xs);
      ymin = fmin(ymin, ys); ymax = fmax(ymax, ys);
    }
    // WAYNE_RNG_SPLIT: the narrow component of a well-populated bin is drawn as one
    // multinomial by k_narrow; k_throw keeps the wide electrons (and whole sparse bins)
    if (a.nsplit) {
      const uint32_t wide = (uint32_t)min(max(nwi, 0), (int32_t)min(c, 0x7FFFFFFFu));
      const uint32_t narrow = c - wide;
      const double sl = a.wa.sigl[w];
      const bool split = a.split_min > 0 && narrow >= (uint32_t)a.split_min && sl > 0.05 &&
                         sl * 6.5 <= (double)kNarrowR;
      // ... and a sparsely populated bin (long scans sampled finely: ~1 electron per bin and
      // sub-sample) is thrown whole by the lane that owns it in k_narrow, from the bin's own
      // Philox blocks: walking such bins electron by electron costs a bin fetch per electron
      const bool sparse = a.split_min > 0 && c > 0 && c < (uint32_t)kSparseMax;
      a.nsplit[(size_t)k * W + w] = split ? (int32_t)narrow : sparse ? -(int32_t)c : 0;
      if (split) { c = wide; n_split_total += narrow; }   // c: electrons left for k_throw
      if (sparse) { n_split_total += c; c = 0; }
    }
  }
  // exclusive scan of c inside the chunk: shuffle scan per wave, wave totals through LDS
  uint32_t incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_wsum[wave] = incl;
  // bounding box of populated bins
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
  if (chunk_total > 0xFFFFFFFFull) overflow = true;
  if (w < W) a.prefix[(size_t)k * (W + 1) + w] = (uint32_t)(wave_off + incl - c);   // chunk-local for now
  if (overflow) atomicExch(a.status, 1);
  // electrons handed to k_narrow: one atomic per workgroup (wave shuffle, then LDS)
  for (int off = 32; off > 0; off >>= 1) n_split_total += __shfl_down(n_split_total, off);
  __shared__ unsigned long long s_split[NW];
  if (lane == 0) s_split[wave] = n_split_total;
  __syncthreads();
  if (tid == 0) {
    unsigned long long tot = 0;
    for (int i = 0; i < NW; ++i) tot += s_split[i];
    if (tot) atomicAdd(a.total_electrons, tot);
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

// One workgroup per sub-sample: chunk offsets -> global exclusive prefix, E_k,
// bounding box -> LDS tile rectangle, SubInfo.
__global__ __launch_bounds__(kPrepThreads) void k_prep_fix(PrepArgs a, int n_chunks) {
  const int k = blockIdx.x;
  const int tid = threadIdx.x;
  const int W = a.W;
  __shared__ uint32_t s_off[64];
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
    if (s_over) atomicExch(a.status, 1);
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
    a.sub[k] = si;
    atomicAdd(a.total_electrons, (unsigned long long)E);
  }
}

// ---------------------------------------------------------------------------
// k_throw : the electron thrower
// ---------------------------------------------------------------------------
// Electrons of sub-sample k are numbered bin-major exactly as the reference
// numbers them (pyparallel_menu.c:87-108) and handed out in units (one RNG
// block of 128 electrons; single electrons in replay mode).  B workgroups
// share the units of sub-sample k evenly, workgroup (k, s) owning a contiguous
// run -- a short slice of the trace, which is what its LDS tile covers.  Inside
// it lane l / wave v takes local slot l*(T/64) + v (T = 512 threads), so the
// 64 lanes of a wave stay spread over the slice: their LDS atomics rarely
// collide in a bank or on a pixel.  Each lane walks its units sequentially and
// re-loads bin parameters (from an LDS copy of the slice's bins) only when it
// crosses a bin boundary.
//
// RNG_MODE 0 (replay): electron i belongs to the emulated OpenMP thread t with
//   t*ssum/T <= i < (t+1)*ssum/T, stream seed 25234 + 17 t + test, and uses
//   rand_r calls 2(i - start_t) and 2(i - start_t)+1 of that stream; the LCG
//   state is reached by an O(log n) affine jump (pyparallel_menu.c:47-61).
//   fp64 Box-Muller, fp64 positions -> bit-exact frames.
// RNG_MODE 1 (Philox): electron e uses words 2j, 2j+1 (j = e mod 128) of the
//   xoshiro128+ stream seeded by Philox block (e / 128, 0, k, exposure), stage
//   STAGE_THROW (philox.h); units are whole blocks, so the draws of an
//   electron do not depend on the launch geometry.  fp32 Box-Muller on the
//   hardware sin/cos/log2 units.
//
// FLUSH 0: add the int32 tile into an int32 frame (wayne_psf_apply).
// FLUSH 1: multiply by the wavelength-dependent flat of THIS sub-sample
//   (grism.py:349-409; applied where the frame is > 0, exposure_generator.py
//   :641-645) and add round(n * flat * 2^28) into the int64 accumulator of the
//   sub-sample's read interval, at the bordered position (y+5, x+5)
//   (detector.py:146-147).  Integer atomics commute, so the result is
//   bit-reproducible for any launch geometry.
constexpr int kThrowThreads = 512;
constexpr int kThrowPCache = 256;       // bins of a workgroup's slice whose prefix / parameters are kept in LDS

struct ThrowArgs {
  int W, K, N, S;          // bins, sub-samples, frame side, bordered side
  int splits;              // workgroups launched per sub-sample (an upper bound: see k_throw)
  int min_wgs;             // spread the electrons over at least this many workgroups per launch
  int threads_compat;      // replay: emulated OpenMP team size
  uint32_t seed, exposure, subsample0;
  uint32_t flags;
  int margin, lds_ints;    // per-workgroup tile: margin around its slice of the trace, LDS capacity
  int flat_off;            // (1014 - N) / 2  (grism.py:363)
  double flat_wmin, flat_wmax, flat_inv_range;   // inv_range = 1 / (wmax - wmin)
  const SubInfo* sub;      // [K]
  const uint32_t* prefix;  // [K*(W+1)]
  const int32_t* nwide;    // [K*W]
  const int32_t* nsplit;   // [K*W] (k_narrow)
  const double* xpos;      // [K*W]
  const double* ypos;      // [K*W]
  const double* sigl;      // [W]
  const double* sigh;      // [W]
  const float* flat[4];    // N*N each or null
  long long* acc;          // FLUSH 1: [R*S*S]
  int32_t* frame;          // FLUSH 0: [N*N]
};

struct Affine { uint32_t a, c; };  // x -> a*x + c (mod 2^32)
__device__ __forceinline__ uint32_t lcg_jump(uint32_t state, uint64_t n) {
  // n steps of next = next*1103515245 + 12345 by square-and-multiply
  uint32_t a = 1103515245u, c = 12345u;   // current power of the map
  uint32_t ra = 1u, rc = 0u;               // accumulated map
  while (n) {
    if (n & 1ull) { ra = ra * a; rc = rc * a + c; }
    c = c * a + c;  // (a,c) o (a,c) = (a*a, a*c + c)
    a = a * a;
    n >>= 1;
  }
  return ra * state + rc;
}
__device__ __forceinline__ int rand_r_step(uint32_t& s) {
  // glibc rand_r: 11 + 10 + 10 bits of three LCG steps
  uint32_t r;
  s = s * 1103515245u + 12345u; r = (s >> 16) & 2047u;
  s = s * 1103515245u + 12345u; r = (r << 10) ^ ((s >> 16) & 1023u);
  s = s * 1103515245u + 12345u; r = (r << 10) ^ ((s >> 16) & 1023u);
  return (int)r;
}

__device__ __forceinline__ double flat_value(const ThrowArgs& a, const SubInfo& si, int x, int y) {
  // grism.py:362-385, evaluated for frame pixel (y, x)
  const int xf = x + a.flat_off, yf = y + a.flat_off;
  const double arr = si.y_ref - (double)yf + si.a_t_i * si.x_ref - si.a_t_i * (double)xf;
  // d = sqrt(arr^2 / (a_t_i^2 + 1)) = |arr| / sqrt(a_t_i^2 + 1); the reciprocals are per
  // sub-sample constants (1 ulp of fp64 from the reference's form, then rounded to float32)
  const double d = fabs(arr) * si.inv_norm;
  const double wl = si.a_w * d + si.b_w;
  const double t = (wl - a.flat_wmin) * a.flat_inv_range;
  const double t2 = t * t, t3 = t2 * t;
  const size_t i = (size_t)y * a.N + x;
  const double f = (double)a.flat[0][i] + ((double)a.flat[1][i] * t) + ((double)a.flat[2][i] * t2) +
                   ((double)a.flat[3][i] * t3);
  // flatfield = np.ones_like(self.flat_f0) is float32, so the assignment
  // rounds the polynomial to float32 (grism.py:380-385)
  return (double)(float)f;
}

template <int FLUSH>
__device__ __forceinline__ void deposit_global(const ThrowArgs& a, const SubInfo& si, int x, int y, int n) {
  if (FLUSH == 0) {
    atomicAdd(&a.frame[(size_t)y * a.N + x], n);
  } else {
    double v = (double)n;
    if ((a.flags & 1u) && a.flat[0]) v = v * flat_value(a, si, x, y);  // WAYNE_F_ADD_FLAT
    const long long q = __double2ll_rn(v * kQ);
    atomicAdd((unsigned long long*)&a.acc[((size_t)si.read * a.S + (y + kBorder)) * a.S + (x + kBorder)],
              (unsigned long long)q);
  }
}

template <int RNG_MODE, int FLUSH>
__global__ __launch_bounds__(kThrowThreads) void k_throw(ThrowArgs a) {
  extern __shared__ int tile[];
  // XCD-aware block -> (sub-sample, split): blocks b and b+8 share an XCD
  // (and its L2); keep all splits of a sub-sample, which read the same
  // prefix / bin arrays and flush to the same frame region, on one XCD.
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int k = (local / a.splits) * 8 + xcd;
  const int s = local % a.splits;
  if (k >= a.K) return;
  const SubInfo si = a.sub[k];
  const uint32_t E = si.electrons;
  if (E == 0) return;
  const int W = a.W;
  const int tid = threadIdx.x;

  const uint32_t* P = a.prefix + (size_t)k * (W + 1);
  const int32_t* NW = a.nwide + (size_t)k * W;
  const double* XP = a.xpos + (size_t)k * W;
  const double* YP = a.ypos + (size_t)k * W;

  // Slots: contiguous electron ranges (whole RNG blocks in Philox mode).  Workgroup s of the
  // sub-sample owns the CONTIGUOUS run of T slots [s T, (s+1) T): a slice of the trace, so its
  // LDS tile only spans that slice plus the PSF margin and few workgroups flush into any pixel.
  // Inside the slice lane l / wave v takes slot l (T/64) + v: the 64 lanes of a wave stay spread.
  constexpr uint32_t UNIT = (RNG_MODE == 1) ? kThrowBlock : 1u;
  const uint64_t n_units = ((uint64_t)E + UNIT - 1) / UNIT;
  // How many of the `splits` launched workgroups share the sub-sample: enough for one unit per lane
  // ("packed": full workgroups, the measured optimum), but at least min_wgs / K so that a few bright
  // sub-samples (staring mode: K = 15) still reach every CU; the host sizes the grid from an estimate
  // of the electrons, so few launched workgroups find themselves beyond B.  The B workgroups take
  // equal shares of the units (floor / ceil), a lane m = ceil(share / T) consecutive units.
  const uint64_t T64 = kThrowThreads;
  uint64_t B = (n_units + T64 - 1) / T64;
  const uint64_t spread = ((uint64_t)a.min_wgs + a.K - 1) / a.K;
  if (B < spread) B = spread;
  if (B > (uint64_t)a.splits) B = (uint64_t)a.splits;
  if (B > n_units) B = n_units;
  if ((uint64_t)s >= B) return;
  const uint64_t u_begin = (uint64_t)s * n_units / B, u_end = ((uint64_t)s + 1) * n_units / B;
  if (u_begin >= u_end) return;
  const uint64_t m_units = (u_end - u_begin + T64 - 1) / T64;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint64_t wg_begin = u_begin * UNIT;
  uint64_t wg_end = u_end * UNIT;
  if (wg_end > E) wg_end = E;

  // First / last bin of the workgroup's electron range, found by all threads at once: thread t owns
  // a chunk of ceil(W/T) bins, the one chunk whose prefix range holds the target finishes the search
  // locally (a per-thread binary search over the whole prefix array costs ~12 dependent HBM/L2 round
  // trips per lane; this costs one round of independent loads plus <= 4 dependent ones in two threads).
  __shared__ int s_rect[4];
  __shared__ int s_bins[2];
  __shared__ uint32_t s_P[kThrowPCache];
  {
    const int c = (W + kThrowThreads - 1) / kThrowThreads;
    const int lo0 = min(tid * c, W), hi0 = min(lo0 + c, W);
    if (lo0 < hi0) {
      const uint32_t plo = P[lo0], phi = P[hi0];
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const uint32_t e = which ? (uint32_t)(wg_end - 1) : (uint32_t)wg_begin;
        if (plo <= e && e < phi) {
          int lo = lo0, hi = hi0;
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P[mid] <= e) lo = mid; else hi = mid; }
          s_bins[which] = lo;
        }
      }
    }
  }
  __syncthreads();
  // (clamped: an inconsistent prefix array must not turn into an out-of-range bin index)
  const int b0 = min(max(s_bins[0], 0), W - 1), b1 = min(max(s_bins[1], b0), W - 1);
  // the slice's prefix entries P[b0 .. b1+1] go to LDS: the per-lane searches below stay on chip
  const int nb = b1 - b0 + 2;
  // (and, for the Philox thrower, the bins' parameters: lanes of a wave cross bin boundaries at
  // different electrons, so nearly every iteration of a wave has some lane fetching a new bin --
  // from LDS that costs ~100 cycles instead of a ~1 us round trip to L2 / HBM)
  const bool p_cached = nb <= kThrowPCache;
  __shared__ float s_par[RNG_MODE == 1 ? 4 * kThrowPCache : 4];
  __shared__ int s_nw[RNG_MODE == 1 ? kThrowPCache : 4];
  if (p_cached) {
    for (int i = tid; i < nb; i += kThrowThreads) s_P[i] = P[b0 + i];
    if (RNG_MODE == 1)
      for (int i = tid; i < nb - 1; i += kThrowThreads) {
        s_par[i] = (float)XP[b0 + i];
        s_par[kThrowPCache + i] = (float)YP[b0 + i];
        s_par[2 * kThrowPCache + i] = (float)a.sigl[b0 + i];
        s_par[3 * kThrowPCache + i] = (float)a.sigh[b0 + i];
        s_nw[i] = max(NW[b0 + i], 0);
      }
  }
  // the workgroup's tile: trace positions of its first and last bin +- margin, clipped to the
  // sub-sample's rectangle (already inside [1, N)) and to the LDS budget
  if (tid == 0) {
    const double xa = fmin(XP[b0], XP[b1]), xb = fmax(XP[b0], XP[b1]);
    const double ya = fmin(YP[b0], YP[b1]), yb = fmax(YP[b0], YP[b1]);
    const double lim = 1e6;
    int x0 = (int)floor(fmax(xa, -lim)) - a.margin, x1 = (int)floor(fmin(xb, lim)) + a.margin + 1;
    int y0 = (int)floor(fmax(ya, -lim)) - a.margin, y1 = (int)floor(fmin(yb, lim)) + a.margin + 1;
    x0 = max(x0, si.tx0); y0 = max(y0, si.ty0);
    x1 = min(x1, si.tx0 + si.tw); y1 = min(y1, si.ty0 + si.th);
    int w_ = max(x1 - x0, 0), h_ = max(y1 - y0, 0);
    while ((long long)w_ * h_ > a.lds_ints && h_ > 1) { y0 += 1; h_ = max(h_ - 2, 1); }
    while ((long long)w_ * h_ > a.lds_ints && w_ > 1) { x0 += 1; w_ = max(w_ - 2, 1); }
    if ((long long)w_ * h_ > a.lds_ints) { w_ = 0; h_ = 0; }
    s_rect[0] = x0; s_rect[1] = y0; s_rect[2] = w_; s_rect[3] = h_;
  }
  __syncthreads();
  const int tx0 = s_rect[0], ty0 = s_rect[1], tw = s_rect[2], th = s_rect[3];
  const int tarea = tw * th;
  for (int i = tid; i < tarea; i += kThrowThreads) tile[i] = 0;
  __syncthreads();

  // inside the workgroup lane l / wave v takes local slot l (T/64) + v: consecutive slots sit in
  // different waves, so a partly filled workgroup still spreads over its 8 waves
  const uint64_t q = (uint64_t)lane * (kThrowThreads / 64) + wave;
  uint64_t ub = u_begin + q * m_units, ue = ub + m_units;
  if (ub > u_end) ub = u_end;
  if (ue > u_end) ue = u_end;
  const uint64_t e_begin64 = ub * UNIT;
  uint64_t e_end64 = ue * UNIT;
  if (e_end64 > E) e_end64 = E;

  if (e_begin64 < e_end64) {
    uint32_t e = (uint32_t)e_begin64;
    const uint32_t e_end = (uint32_t)e_end64;
    // bin b with P[b] <= e < P[b+1], inside the workgroup's [b0, b1]
    int b;
    uint32_t bin_start, bin_end;
    if (p_cached) {
      int lo = 0, hi = nb - 1;   // invariant: s_P[lo] <= e < s_P[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_P[mid] <= e) lo = mid; else hi = mid;
      }
      b = b0 + lo; bin_start = s_P[lo]; bin_end = s_P[lo + 1];
    } else {
      int lo = b0, hi = b1 + 1;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (P[mid] <= e) lo = mid; else hi = mid;
      }
      b = lo; bin_start = P[b]; bin_end = P[b + 1];
    }
    uint32_t wide_end = (RNG_MODE == 1 && p_cached) ? 0u : bin_start + (uint32_t)max(NW[b], 0);

    if (RNG_MODE == 1) {
      float x, y, sl, sh;
      if (p_cached) {
        const int i = b - b0;
        x = s_par[i]; y = s_par[kThrowPCache + i]; sl = s_par[2 * kThrowPCache + i]; sh = s_par[3 * kThrowPCache + i];
        wide_end = bin_start + (uint32_t)s_nw[i];
      } else {
        x = (float)XP[b]; y = (float)YP[b];
        sl = (float)a.sigl[b]; sh = (float)a.sigh[b];
      }
      while (e < e_end) {
        // one seeded stream per block of kThrowBlock electrons
        SeededStream rng(a.seed, STAGE_THROW, e / kThrowBlock, (uint32_t)k + a.subsample0, a.exposure);
        const uint32_t blk_end = min(e_end, (e / kThrowBlock + 1u) * kThrowBlock);
        for (; e < blk_end; ++e) {
          if (e >= bin_end) {
            if (p_cached) {
              int i = b - b0;
              do { ++i; bin_start = bin_end; bin_end = s_P[i + 1]; } while (bin_end <= e && i + 2 < nb);
              b = b0 + i;
              x = s_par[i]; y = s_par[kThrowPCache + i]; sl = s_par[2 * kThrowPCache + i]; sh = s_par[3 * kThrowPCache + i];
              wide_end = bin_start + (uint32_t)s_nw[i];
            } else {
              do { ++b; bin_start = bin_end; bin_end = P[b + 1]; } while (bin_end <= e && b + 1 < W);
              wide_end = bin_start + (uint32_t)max(NW[b], 0);
              x = (float)XP[b]; y = (float)YP[b];
              sl = (float)a.sigl[b]; sh = (float)a.sigh[b];
            }
          }
          const float ua = u01f(rng.next());
          const float ub = u01f(rng.next());
          // R = sqrt(-2 ln ub) = sqrt(-2 ln2 log2 ub); sin/cos take revolutions
          const float R = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(ub));
          const float zx = R * __builtin_amdgcn_cosf(ua);
          const float zy = R * __builtin_amdgcn_sinf(ua);
          const float sig = (e < wide_end) ? sh : sl;   // first N electrons: wide gaussian (:89-98)
          const int xi = (int)fmaf(zx, sig, x);          // C truncation toward zero (:91-92)
          const int yi = (int)fmaf(zy, sig, y);
          const int lx = xi - tx0, ly = yi - ty0;
          // the tile lies inside [1, N) x [1, N), so this one test implies the
          // reference's 0 < pos < n bounds (:93) on the fast path
          if ((unsigned)lx < (unsigned)tw && (unsigned)ly < (unsigned)th)
            atomicAdd(&tile[__umul24(ly, tw) + lx], 1);   // tile area < 2^14: 24-bit multiply-add
          else if (xi > 0 && xi < a.N && yi > 0 && yi < a.N)
            deposit_global<FLUSH>(a, si, xi, yi, 1);
        }
      }
    } else {
      // replay: electron i belongs to the emulated OpenMP thread t that owns
      // [t*E/T, (t+1)*E/T) (the last one ends at E, :48-49)
      const int T = a.threads_compat;
      auto part_start_of = [&](int t) -> uint32_t { return (uint32_t)(((long long)t * (long long)E) / T); };
      int part = (int)(((unsigned long long)e * (unsigned long long)T) / E);
      if (part >= T) part = T - 1;
      while (part > 0 && part_start_of(part) > e) --part;
      while (part + 1 < T && part_start_of(part + 1) <= e) ++part;
      uint32_t part_end = (part == T - 1) ? E : part_start_of(part + 1);
      uint32_t lcg = lcg_jump((uint32_t)(25234 + 17 * part + si.replay_seed), 6ull * (uint64_t)(e - part_start_of(part)));
      double x = XP[b], y = YP[b];
      double sl = a.sigl[b], sh = a.sigh[b];
      for (; e < e_end; ++e) {
        if (e >= bin_end) {
          do { ++b; bin_start = bin_end; bin_end = P[b + 1]; } while (bin_end <= e && b + 1 < W);
          wide_end = bin_start + (uint32_t)max(NW[b], 0);
          x = XP[b]; y = YP[b]; sl = a.sigl[b]; sh = a.sigh[b];
        }
        while (e >= part_end && part + 1 < T) {   // next emulated thread: fresh stream
          ++part;
          part_end = (part == T - 1) ? E : part_start_of(part + 1);
          lcg = (uint32_t)(25234 + 17 * part + si.replay_seed);
        }
        // pyparallel_menu.c:57-61
        const double theta = 2. * kPi * rand_r_step(lcg) / ((double)2147483647);
        const double R = sqrt(-2. * log(rand_r_step(lcg) / ((double)2147483647)));
        const double zx = R * cos(theta);
        const double zy = R * sin(theta);
        const double sig = (e < wide_end) ? sh : sl;
        const double px = zx * sig + x, py = zy * sig + y;
        // (int) of a non-finite / out-of-range double: reject (x86 gives INT_MIN)
        const bool okx = (px > -2147483649.0 && px < 2147483648.0);
        const bool oky = (py > -2147483649.0 && py < 2147483648.0);
        const int xi = okx ? (int)px : -1, yi = oky ? (int)py : -1;
        const int lx = xi - tx0, ly = yi - ty0;
        if ((unsigned)lx < (unsigned)tw && (unsigned)ly < (unsigned)th)
          atomicAdd(&tile[ly * tw + lx], 1);
        else if (xi > 0 && xi < a.N && yi > 0 && yi < a.N)
          deposit_global<FLUSH>(a, si, xi, yi, 1);
      }
    }
  }
  __syncthreads();
  // flush the tile
  for (int i = tid; i < tarea; i += kThrowThreads) {
    const int n = tile[i];
    if (n > 0) {
      const int ly = i / tw, lx = i - ly * tw;
      deposit_global<FLUSH>(a, si, tx0 + lx, ty0 + ly, n);
    }
  }
}

// ---------------------------------------------------------------------------
// k_narrow : the narrow PSF component of a bin as ONE multinomial draw
// ---------------------------------------------------------------------------
// Throwing n electrons independently at pixels with probabilities p_ij is the
// multinomial(n; p_ij) distribution of the pixel counts.  For the narrow
// gaussian (sigma_l = 0.5-0.9 px, ~80 % of the electrons, pyparallel_menu.c:99-107)
// nearly all of the mass sits in a 5 x 5 block, so the counts are drawn
// directly: x and y are independent, so first the column counts (a chain of
// conditional binomials, centre column outwards), then each non-empty column's
// row counts.  Cell probabilities are differences of gaussian upper tails
// (pixel i holds positions [i, i+1): the reference's (int) truncation, which is
// floor() wherever a pixel is kept, :91-93).  ~30-40 binomial draws replace
// ~1400 electron throws per bin; the distribution of the frame is the same.
//
// One lane per bin, 256 consecutive bins per workgroup, cells visited in
// lockstep with wave-level skipping (a cell is processed only while some lane
// still holds electrons).  Random words: the bin's STAGE_NARROW stream.
constexpr int kNarrowThreads = 256;
constexpr int kNarrowCells = 2 * kNarrowR + 1;
constexpr int kNarrowTile = 1536;       // ints of LDS for the workgroup's tile (its bins span ~15 x 1 px + the 13 x 13 windows)

__device__ __forceinline__ float upper_tail(float t) { return 0.5f * erfcf(t * 0.70710678118654752f); }

template <int FLUSH, bool FAST>
__global__ __launch_bounds__(kNarrowThreads) void k_narrow(ThrowArgs a) {
  typedef typename std::conditional<FAST, FastMath, ExactMath<float> >::type M;
  __shared__ int tile[kNarrowTile];
  __shared__ float s_q[kNarrowCells][kNarrowThreads];   // row probabilities of each lane's bin
  __shared__ int s_box[4];
  const int k = blockIdx.y;
  const int tid = threadIdx.x;
  const int w = blockIdx.x * kNarrowThreads + tid;
  const SubInfo si = a.sub[k];
  const int n0 = (w < a.W) ? a.nsplit[(size_t)k * a.W + w] : 0;   // > 0 multinomial, < 0 sparse bin
  if (!__syncthreads_or(n0 != 0)) return;

  float x = 0.f, y = 0.f, sg = 1.f;
  int ic0 = 0, jc0 = 0;
  if (n0 != 0) {
    x = (float)a.xpos[(size_t)k * a.W + w];
    y = (float)a.ypos[(size_t)k * a.W + w];
    sg = (float)a.sigl[w];
    ic0 = (int)floorf(x);
    jc0 = (int)floorf(y);
  }
  // workgroup tile = bounding box of its bins' windows, clipped to [1, N)
  if (tid == 0) { s_box[0] = 0x7FFFFFFF; s_box[1] = -0x7FFFFFFF; s_box[2] = 0x7FFFFFFF; s_box[3] = -0x7FFFFFFF; }
  __syncthreads();
  if (n0 != 0) {
    atomicMin(&s_box[0], ic0 - kNarrowR); atomicMax(&s_box[1], ic0 + kNarrowR + 1);
    atomicMin(&s_box[2], jc0 - kNarrowR); atomicMax(&s_box[3], jc0 + kNarrowR + 1);
  }
  __syncthreads();
  int tx0 = max(s_box[0], 1), tx1 = min(s_box[1], a.N), ty0 = max(s_box[2], 1), ty1 = min(s_box[3], a.N);
  int tw = max(tx1 - tx0, 0), th = max(ty1 - ty0, 0);
  if ((long long)tw * th > kNarrowTile) { th = min(th, kNarrowTile / max(tw, 1)); if (th < 1) { th = 0; tw = 0; } }
  const int tarea = tw * th;
  for (int i = tid; i < tarea; i += kNarrowThreads) tile[i] = 0;

  // (waves that hold only sparse or empty bins skip the multinomial altogether)
  const bool any_multi = __any(n0 > 0);
  const float inv_s = 1.f / sg;
  if (any_multi) {
    // row probabilities, centre-out: c = 0 centre, odd c -> +((c+1)/2), even c -> -(c/2)
    {
      const float f = y - (float)jc0;
      float up = upper_tail((1.f - f) * inv_s), lo = upper_tail(f * inv_s);   // mass above / below the centre row
      s_q[0][tid] = 1.f - up - lo;
      for (int c = 1; c < kNarrowCells; ++c) {
        const int d = (c + 1) >> 1;
        if (c & 1) { const float nx = upper_tail(((float)(d + 1) - f) * inv_s); s_q[c][tid] = up - nx; up = nx; }
        else       { const float nx = upper_tail(((float)d + f) * inv_s);       s_q[c][tid] = lo - nx; lo = nx; }
      }
    }
  }
  __syncthreads();

  if (any_multi) {
    SeededStream rng(a.seed, STAGE_NARROW, (uint32_t)w, (uint32_t)k + a.subsample0, a.exposure);
    const float fx = x - (float)ic0;
    float up = upper_tail((1.f - fx) * inv_s), lo = upper_tail(fx * inv_s);
    float n_rem = (float)max(n0, 0);
    for (int c = 0; c < kNarrowCells; ++c) {
      if (!__any(n_rem > 0.f)) break;
      // this column's mass and the mass of everything not yet visited (before it)
      const int d = (c + 1) >> 1;
      const int ci = (c == 0) ? ic0 : ((c & 1) ? ic0 + d : ic0 - d);
      float P, rem;
      if (c == 0) { P = 1.f - up - lo; rem = 1.f; }
      else if (c & 1) { const float nx = upper_tail(((float)(d + 1) - fx) * inv_s); P = up - nx; rem = up + lo; up = nx; }
      else            { const float nx = upper_tail(((float)d + fx) * inv_s);       P = lo - nx; rem = up + lo; lo = nx; }
      float n_col = 0.f;
      if (n_rem > 0.f) {
        const float pc = fminf(fmaxf(M::div_(P, rem), 0.f), 1.f);
        n_col = binomial<M>(n_rem, pc, rng);
        n_rem -= n_col;
      }
      if (!__any(n_col > 0.f)) continue;
      // rows of this column
      float m_rem = n_col, qrem = 1.f;
      for (int r = 0; r < kNarrowCells; ++r) {
        if (!__any(m_rem > 0.f)) break;
        const float Q = s_q[r][tid];
        float m = 0.f;
        if (m_rem > 0.f) {
          const float qc = fminf(fmaxf(M::div_(Q, qrem), 0.f), 1.f);
          m = binomial<M>(m_rem, qc, rng);
          m_rem -= m;
        }
        qrem -= Q;
        if (m > 0.f) {
          const int e = (r + 1) >> 1;
          const int rj = (r == 0) ? jc0 : ((r & 1) ? jc0 + e : jc0 - e);
          const int lx = ci - tx0, ly = rj - ty0;
          if ((unsigned)lx < (unsigned)tw && (unsigned)ly < (unsigned)th)
            atomicAdd(&tile[ly * tw + lx], (int)m);
          else if (ci > 0 && ci < a.N && rj > 0 && rj < a.N)       // (:93)
            deposit_global<FLUSH>(a, si, ci, rj, (int)m);
        }
      }
    }
  }
  // sparse bins: electron j of the bin takes words 2(j&1), 2(j&1)+1 of Philox block (w, j/2, k, exposure),
  // stage STAGE_SPARSE; the first n_wide electrons get sigma_h as everywhere (pyparallel_menu.c:89-107)
  {
    const int cs = (n0 < 0) ? -n0 : 0;
    int nw = 0;
    float sh = 1.f;
    if (cs > 0) { nw = max(a.nwide[(size_t)k * a.W + w], 0); sh = (float)a.sigh[w]; }
    for (int j = 0; __any(j < cs); j += 2) {
      if (j < cs) {
        const u32x4 r = philox4x32_10((uint32_t)w, (uint32_t)(j >> 1), (uint32_t)k + a.subsample0, a.exposure,
                                      a.seed, STAGE_SPARSE);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (j + h < cs) {
            const float ua = u01f(r.v[2 * h]), ub = u01f(r.v[2 * h + 1]);
            const float R = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(ub));
            const float sig = (j + h < nw) ? sh : sg;
            const int xi = (int)fmaf(R * __builtin_amdgcn_cosf(ua), sig, x);
            const int yi = (int)fmaf(R * __builtin_amdgcn_sinf(ua), sig, y);
            const int lx = xi - tx0, ly = yi - ty0;
            if ((unsigned)lx < (unsigned)tw && (unsigned)ly < (unsigned)th)
              atomicAdd(&tile[ly * tw + lx], 1);
            else if (xi > 0 && xi < a.N && yi > 0 && yi < a.N)
              deposit_global<FLUSH>(a, si, xi, yi, 1);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < tarea; i += kNarrowThreads) {
    const int n = tile[i];
    if (n > 0) {
      const int ly = i / tw, lx = i - ly * tw;
      deposit_global<FLUSH>(a, si, tx0 + lx, ty0 + ly, n);
    }
  }
}

// ---------------------------------------------------------------------------
// k_cosmic : MinMaxPossionCosmicGenerator.cosmic_frame (cosmic_rays.py:70-139)
// ---------------------------------------------------------------------------
struct CosmicArgs {
  int R, N, S;
  uint32_t seed, exposure;
  double rate;               // hits per second per 1024^2 pixels
  const double* read_dt;     // [R]
  long long* acc;            // [R*S*S]
};

__global__ __launch_bounds__(256) void k_cosmic(CosmicArgs a) {
  const int r = blockIdx.x;
  if (r >= a.R) return;
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) {
    // rate_size = rate / (1024*1024) * N*N ; Poisson(rate_size * time)  (:33-44, :121-127)
    const double rate_size = a.rate / (1024. * 1024.) * (double)((long long)a.N * a.N);
    PhiloxStream rng(a.seed, STAGE_CR_COUNT, 0u, (uint32_t)r, a.exposure);
    double n = poisson<ExactMath<double> >(rate_size * a.read_dt[r], rng);
    if (!(n >= 0.)) n = 0.;
    if (n > 1e7) n = 1e7;
    s_n = (uint32_t)n;
  }
  __syncthreads();
  const uint32_t n = s_n;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const u32x4 w = philox4x32_10(i, 0u, (uint32_t)r, a.exposure, a.seed, STAGE_CR_HIT);
    const uint32_t energy = 10000u + uint_below(w.v[0], 25000u);  // randint(10000, 35000)  (:134)
    const uint32_t y = uint_below(w.v[1], (uint32_t)a.N);          // randint(0, len(array))  (:80)
    const uint32_t x = uint_below(w.v[2], (uint32_t)a.N);          // randint(0, len(array[0])) (:81)
    const long long q = (long long)energy << kQBits;
    atomicAdd((unsigned long long*)&a.acc[((size_t)r * a.S + (y + kBorder)) * a.S + (x + kBorder)],
              (unsigned long long)q);
  }
}

// ---------------------------------------------------------------------------
// k_ramp : fused up-the-ramp kernel, one thread per bordered pixel
// ---------------------------------------------------------------------------
struct RampArgs {
  int R, N, S;
  uint32_t seed, exposure, flags;
  double sky_ct_s;           // <= 0: no sky
  double noise_mean, noise_std;
  const double* read_dt;     // [R]
  long long* acc;            // [R*S*S] read and cleared
  const float* pfl;          // [S*S] (bordered layout; border unused) or null
  const float* sky;          // [S*S] or null
  const float* lin[4];       // [S*S] or null
  const float* dark_sci;     // [R*S*S] or null
  const float* dark_err;
  const double* zero_read;   // [S*S] or null
  void* out;                 // [(R+1)*S*S] float or double
  // sky background (see sky_draw): alias tables of Poisson(level_j * bg_count) for `sky_levels` levels
  // of the master sky and every distinct read interval, for the reads whose bit is set in alias_mask
  const uint32_t* sky_alias; // [n_tables <= kMaxReads][kSkyAlias] or null
  uint32_t alias_mask;
  int sky_levels;            // L
  float sky_level[16];       // ascending levels of the master sky (quantiles of its positive pixels; [0] = min)
  unsigned char sky_tab0[16];  // first table of read r (its level-0 table; level j is the j-th after it)
};

constexpr int kSkyAlias = 256;   // entries per alias table: alias << 24 | 24-bit acceptance threshold
constexpr float kSkyPiece = 16.f; // largest mean drawn by one sequential search

constexpr int kRampThreads = 256;
constexpr int kMaxReads = 15;   // NSAMP <= 16 (detector.py:228)

// Box-Muller pair from two words.  EXACT mirrors the oracle's libm formula;
// FAST uses the hardware units (sin / cos take revolutions, log is log2).
template <bool FAST>
__device__ __forceinline__ void bm_pair(uint32_t w0, uint32_t w1, float& z0, float& z1) {
  const float ua = u01f(w0), ub = u01f(w1);
  if (FAST) {
    const float Rr = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(ub));
    z0 = Rr * __builtin_amdgcn_cosf(ua);
    z1 = Rr * __builtin_amdgcn_sinf(ua);
  } else {
    const float Rr = sqrtf(-2.0f * logf(ub));
    const float ang = 6.283185307179586f * ua;
    z0 = Rr * cosf(ang);
    z1 = Rr * sinf(ang);
  }
}

__device__ __forceinline__ double nonlinear_response(double px, float c1, float c2, float c3, float c4) {
  // WFC3_IR.apply_non_linearity (detector.py:335-348): Newton-Raphson on
  // u (1 + c1 + u (c2 + u (c3 + c4 u))) = px from u0 = px, until |du| < 1e-3.
  // The reference iterates the whole frame until its slowest pixel converges;
  // here each pixel stops on its own criterion (the extra iterations move a
  // converged pixel by < 1e-9).  (1 + c1), 2*c2, 3*c3, 4*c4 are float32 in
  // the reference because the coefficient planes are.  The residual is
  // evaluated in fp64; the reciprocal of the derivative (within 2e-7 of
  // 1 + small) in fp32, which changes an iterate by < 1e-7 of its step.
  const double k1 = (double)(1.0f + c1);
  const double k2 = (double)c2, k3 = (double)c3, k4 = (double)c4;
  const float d1 = 1.0f + c1, d2 = 2.0f * c2, d3 = 3.0f * c3, d4 = 4.0f * c4;
  double u0 = px, u1 = px;
  for (int it = 0; it < 10000; ++it) {
    const double f = fma(u0, fma(u0, fma(u0, fma(k4, u0, k3), k2), k1), -px);
    const float uf = (float)u0;
    const float fp_ = fmaf(uf, fmaf(uf, fmaf(uf, d4, d3), d2), d1);
    const double step = f * (double)__builtin_amdgcn_rcpf(fp_);
    u1 = u0 - step;
    if (fabs(step) < 1e-3) break;
    u0 = u1;
  }
  return u1;
}

// Sky background of one read interval (exposure_generator.py:488-495: pixel += poisson(master_sky *
// bg_count)).  All pixels of the frame share bg_count and the master sky is flat to a few per cent, so
// the draw is split with the additivity of Poisson variables:
//     Poisson(sky_px * bg) = Poisson(level_j * bg) + Poisson((sky_px - level_j) * bg),
// level_j the highest of L levels (quantiles of the master sky) not above sky_px.  The first term comes from an
// alias table (Walker / Vose) shared by every pixel of that level -- one random word, one LDS read; the
// second has a mean of a fraction of an electron to a few electrons and is drawn by inversion from 0
// (one word, a two- or three-step search).  No rejection loop, hardly any divergence: ~50
// instructions instead of ~180 for a transformed-rejection draw per pixel, and exactly Poisson.
// Exposures with a read whose rate does not fit the table (long reads under a bright sky) take the
// ALIAS = false variant of k_ramp: Poisson(lam) per pixel by Knuth / PTRS (sky_counts below).
template <class M, bool PIECES, class RNG>
__device__ __forceinline__ float sky_draw(const uint32_t* tab, float lam_level, float lam, RNG& rng) {
  // shared part: N ~ Poisson(lam_level)
  const uint32_t w = rng.next();
  const uint32_t idx = w >> 24;
  const uint32_t e = tab[idx];
  float k = (float)(((w & 0xFFFFFFu) < (e & 0xFFFFFFu)) ? idx : (e >> 24));
  // the pixel's own part: Poisson(lam - lam_level) by sequential search from 0.  PIECES (chosen by the
  // host when some pixel of the master sky lies far above its level -- a hot pixel): in pieces of mean
  // <= kSkyPiece (additivity again), so that exp(-mean) stays far from underflow whatever the plane holds
  float ld = lam - lam_level;
  if (ld > 0.f) {
    for (;;) {
      const float piece = PIECES ? fminf(ld, kSkyPiece) : ld;
      float u = M::u01(rng.next());
      float pk = M::exp_(-piece);
      float j = 0.f;
      for (int it = 0; it < 512; ++it) {
        if (u <= pk) break;
        u = u - pk;
        j = j + 1.f;
        pk = pk * M::div_(piece, j);
      }
      k = k + j;
      if (!PIECES) break;
      ld = ld - piece;
      if (!(ld > 0.f)) break;
    }
  }
  return k;
}

// Phase 1 of k_ramp: the sky Poisson draws of one pixel for all reads
// (exposure_generator.py:488-495), from the pixel's seeded stream, written to
// LDS.  Lanes advance through their reads independently ("lane-asynchronous"):
// a lane whose trial is rejected retries while its neighbours move on to their
// next read, so a wave runs ~R * 1.15 trial rounds instead of R * (the slowest
// of 64 lanes).  The per-pixel draw sequence is sequential, so the result does
// not depend on this scheduling.
template <bool FAST>
__device__ __forceinline__ void sky_counts(const RampArgs& a, uint32_t p, int tid, bool active, float skyv,
                                           const float* s_c, uint32_t (*s_sky)[kRampThreads]) {
  typedef typename std::conditional<FAST, FastMath, ExactMath<float> >::type M;
  SeededStream rng(a.seed, STAGE_SKY, p, 0u, a.exposure);
  int r = 0, mode = 0, guard = 0;
  PtrsSetup<M> ps;
  ps.lam = ps.b = ps.a = ps.vr = ps.loglam = ps.invalpha = 0.f;
  float prod = 1.f, enlam = 0.f, kk = 0.f, lam = 0.f;
  const int R = a.R;
  while (__any(active)) {
    if (active) {
      float done = -1.f;
      if (mode == 0) {
        // master_sky *= bg_count is an in-place float32 multiply (:493)
        lam = skyv * s_c[r];
        if (!(lam > 0.f)) {
          done = 0.f;
        } else if (lam < 10.f) {
          enlam = M::exp_(-lam); prod = 1.f; kk = 0.f; mode = 1;
        } else if (lam < 256.f) {
          if (lam != ps.lam) ps.init(lam);   // SPARS / STEP sequences repeat their read interval: same lam again
          mode = 2;
        } else {
          done = (float)poisson<ExactMath<double> >((double)lam, rng);   // rare: long reads
        }
      }
      if (mode == 1) {
        prod = prod * M::u01(rng.next());
        if (prod > enlam) kk = kk + 1.f; else done = kk;
      } else if (mode == 2) {
        const uint32_t w1 = rng.next();
        const uint32_t w2 = rng.next();
        float k;
        if (ps.trial(w1, w2, k)) done = k;
      }
      if (++guard > 64 + 600 * kMaxReads && done < 0.f) done = floorf(lam + 0.5f);   // unreachable safety net
      if (done >= 0.f) {
        s_sky[r][tid] = (uint32_t)done;
        mode = 0;
        if (++r >= R) active = false;
      }
    }
  }
}

// SKY: 0 = Poisson(lam) per pixel (sky_counts), 1 = alias tables + one-piece remainder, 2 = alias tables +
// remainder in pieces (a master sky with hot pixels)
template <class OutT, bool FAST, int SKY>
__global__ __launch_bounds__(kRampThreads) void k_ramp(RampArgs a) {
  constexpr bool ALIAS = SKY != 0;
  typedef typename std::conditional<FAST, FastMath, ExactMath<float> >::type M;
  static_assert(kSkyAlias == kRampThreads, "the sky tables and the per-thread sky counts share one LDS array");
  __shared__ uint32_t s_tab[kMaxReads][kSkyAlias];   // ALIAS: alias tables; else: sky counts [read][thread]
  __shared__ float s_c[kMaxReads + 1];
  const int S = a.S;
  const int tid = threadIdx.x;
  const int p_raw = blockIdx.x * blockDim.x + tid;
  const bool valid = p_raw < S * S;
  const int p = valid ? p_raw : 0;
  const int Y = p / S, X = p - Y * S;
  const bool interior = valid && (X >= kBorder && X < S - kBorder && Y >= kBorder && Y < S - kBorder);
  const size_t SS = (size_t)S * S;
  OutT* out = (OutT*)a.out;
  const bool clip = (a.flags & (1u << 3)) != 0;
  const bool rdn = (a.flags & (1u << 4)) != 0;
  const bool do_dark = (a.flags & (1u << 6)) != 0 && a.dark_sci && a.dark_err;
  const bool do_lin = (a.flags & (1u << 2)) != 0 && a.lin[0];
  const bool gainvar = (a.flags & (1u << 1)) != 0 && a.pfl;
  const bool do_noise = (a.noise_mean != 0.) && (a.noise_std != 0.);   // `if noise_mean and noise_std` (:477)
  const bool do_sky = a.sky_ct_s > 0. && a.sky;

  if (tid < a.R) s_c[tid] = (float)(a.sky_ct_s * a.read_dt[tid]);        // bg_count of read tid (:489-491)
  if (ALIAS && do_sky)
    for (int i = tid; i < kMaxReads * kSkyAlias; i += kRampThreads) (&s_tab[0][0])[i] = a.sky_alias[i];
  float skyv = 0.f;
  if (interior && do_sky) skyv = a.sky[p];
  __syncthreads();
  if (!ALIAS) {
    if (do_sky) sky_counts<FAST>(a, (uint32_t)p, tid, interior && skyv > 0.f, skyv, s_c, s_tab);
    __syncthreads();
  }
  if (!valid) return;
  // the pixel's sky level (constant over the reads): the highest level not above its sky value
  int sky_lvl = 0;
  if (ALIAS && skyv > 0.f)
    for (int l = 1; l < a.sky_levels; ++l) sky_lvl += (a.sky_level[l] <= skyv) ? 1 : 0;
  const float sky_base = a.sky_level[sky_lvl];

  // per-pixel streams, seeded only when the stage is on (one Philox block each)
  SeededStream rn, rg, rs;
  if (ALIAS && skyv > 0.f) rs = SeededStream(a.seed, STAGE_SKY, (uint32_t)p, 0u, a.exposure);
  if (rdn || do_dark) rn = SeededStream(a.seed, STAGE_READ, (uint32_t)p, 0u, a.exposure);
  if (do_noise) rg = SeededStream(a.seed, STAGE_NOISE, (uint32_t)p, 0u, a.exposure);

  // zero read: (initial bias) -> clip -> reference pixels := 0 -> read noise
  // (exposure_generator.py:446-466, exposure.py:82-131, 61-68)
  double z = (a.zero_read && (a.flags & (1u << 7))) ? a.zero_read[p] : 0.;
  if (clip) z = fmin(fmax(z, kMinCounts), kMaxCounts);
  if (!interior) z = 0.;
  {
    float zd, zr;
    const uint32_t w0 = rn.next(), w1 = rn.next();
    double v = z;
    if (rdn) { bm_pair<FAST>(w0, w1, zd, zr); v = v + kReadNoise * (double)zr; }
    out[p] = (OutT)v;
  }

  // gain: 2.35 / pfl evaluated in float32 as numpy does for scalar / f32 array
  // (detector.py:203-204), or the constant (exposure_generator.py:507-511);
  // applied as a multiplication by its fp64 reciprocal
  double inv_g = 1.0 / kGain;
  float c1 = 0, c2 = 0, c3 = 0, c4 = 0;
  if (interior && gainvar) inv_g = 1.0 / (double)(2.35f / a.pfl[p]);
  if (do_lin) { c1 = a.lin[0][p]; c2 = a.lin[1][p]; c3 = a.lin[2][p]; c4 = a.lin[3][p]; }

  // software-pipelined ramp: the planes of read r+1 are requested before the
  // (VALU-heavy) work on read r so that HBM latency hides behind it
  long long* __restrict__ accp = a.acc + p;
  const float* __restrict__ dsp = a.dark_sci ? a.dark_sci + p : nullptr;
  const float* __restrict__ dep = a.dark_err ? a.dark_err + p : nullptr;
  const bool ld_dark = do_dark && interior;
  long long q_next = interior ? accp[0] : 0;
  float ds_next = ld_dark ? dsp[0] : 0.f, de_next = ld_dark ? dep[0] : 0.f;
  double cum = 0.;
  for (int r = 0; r < a.R; ++r) {
    const long long q = q_next;
    const float ds = ds_next, de = de_next;
    if (r + 1 < a.R) {
      if (interior) q_next = accp[(size_t)(r + 1) * SS];
      if (ld_dark) { ds_next = dsp[(size_t)(r + 1) * SS]; de_next = dep[(size_t)(r + 1) * SS]; }
    }
    double px = 0.;
    const uint32_t g0 = do_noise ? rg.next() : 0u, g1 = do_noise ? rg.next() : 0u;
    if (interior) {
      accp[(size_t)r * SS] = 0;      // leave the accumulator clean for the next exposure
      px = (double)q * kInvQ;
      if (do_noise) {                // _gen_noise (:477-484, :712-727)
        const double dt = a.read_dt[r];
        float z0, z1;
        bm_pair<FAST>(g0, g1, z0, z1);
        px = px + (a.noise_mean * dt + (a.noise_std * dt) * (double)z0);
      }
      if (skyv > 0.f) {                                  // += np.random.poisson(master_sky) (:495)
        // master_sky *= bg_count is an in-place float32 multiply (:493)
        const float lam = skyv * s_c[r];
        if (ALIAS) {
          if (lam > 0.f) px = px + (double)sky_draw<M, SKY == 2>(s_tab[a.sky_tab0[r] + sky_lvl], sky_base * s_c[r], lam, rs);
        } else {
          px = px + (double)s_tab[r][tid];
        }
      }
      px = px * inv_g;               // electrons -> DN (:507-511)
    }
    cum = cum + px;                  // cumulative_pixel_array += pixel_array_full (:378)
    double v = cum;
    float zd = 0.f, zr = 0.f;
    const uint32_t w0 = rn.next(), w1 = rn.next();
    if (rdn || ld_dark) bm_pair<FAST>(w0, w1, zd, zr);
    if (interior) {
      if (do_dark) {                 // detector.py:185-191
        const double err = (de > 0.f) ? (double)de : (double)0.00001f;
        v = v + ((double)ds + err * (double)zd);
      }
      if (do_lin) v = nonlinear_response(v, c1, c2, c3, c4);
      if (clip) v = fmin(fmax(v, kMinCounts), kMaxCounts);
    } else {
      v = 0.;                        // reset_reference_pixels (exposure.py:122-131)
    }
    v = v + z;                       // add_zero_read (exposure.py:94-104)
    if (rdn) v = v + kReadNoise * (double)zr;   // add_read_noise (detector.py:193-198)
    out[(size_t)(r + 1) * SS + p] = (OutT)v;
  }
}

}  // namespace wayne
