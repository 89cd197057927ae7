// k_throw: the electron thrower (A1-A4, A11, A12)
#pragma once
#include "common.h"
#include "k_prep.h"

namespace wayne {

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
// (kMaxChunks: plan_consts.h)
constexpr int kThrowPCache = 256;       // bins of a workgroup's slice whose prefix / parameters are kept in LDS

struct ThrowArgs {
  int W, K, N, S;          // bins, sub-samples, frame side, bordered side
  int kb;                  // k_lane: consecutive sub-samples a workgroup takes (>= 1; see k_lane, "BATCHES")
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
  const int32_t* nlane;    // [K*W] (k_lane)
  // k_narrow / k_lane: chunk (of one workgroup's bins) handled by the workgroups with blockIdx.y = rank -- heaviest chunks
  // first, so that the last workgroups of the launch, which run on a nearly empty chip, are the light ones
  unsigned char chunk_order[kMaxChunks];
  unsigned char lane_order[kMaxChunks];    // the same for k_lane's chunks (kLaneThreads bins each)
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

// What the per-electron loop needs of a bin, as one 32-byte record (two LDS reads)
struct ThrowBin {
  float x, y;             // trace position: the fraction of its pixel (common.h, bin_local) ...
  float ch, cl;           // -2 ln2 sigma_h^2, -2 ln2 sigma_l^2
  uint32_t bin_end;       // first electron (sub-sample numbering) after this bin
  uint32_t wide_end;      // first electron of the bin that takes sigma_l
  int ox, oy;             // ... and that pixel (frame coordinates)
};

__device__ __forceinline__ ThrowBin load_throw_bin(const uint32_t* P, const int32_t* NW, const double* XP, const double* YP,
                                                   const double* sigl, const double* sigh, int b) {
  ThrowBin r;
  const BinLocal bl = bin_local(XP[b], YP[b]);
  r.x = bl.fx; r.y = bl.fy; r.ox = bl.ox; r.oy = bl.oy;
  const float sh = (float)sigh[b], sl = (float)sigl[b];
  r.ch = (-1.3862943611198906f * sh) * sh;
  r.cl = (-1.3862943611198906f * sl) * sl;
  const uint32_t start = P[b];
  r.bin_end = P[b + 1];
  // N = (int)(counts * ratio) may exceed the bin's count (ratio > 1: every electron wide, pyparallel_menu.c:89-98)
  r.wide_end = start + min((uint32_t)max(NW[b], 0), r.bin_end - start);
  return r;
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
  // Fewer than 8 sub-samples (wayne_psf_apply: one): that rule would leave XCDs idle -- the one sub-sample of a
  // psf_apply call ran on an eighth of the chip -- so the splits go round the XCDs instead (grid = K * splits).
  int k, s;
  if (a.K < 8) {
    k = (int)(blockIdx.x % (unsigned)a.K);
    s = (int)(blockIdx.x / (unsigned)a.K);
  } else {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    k = (local / a.splits) * 8 + xcd;
    s = local % a.splits;
  }
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
  __shared__ ThrowBin s_bin[RNG_MODE == 1 ? kThrowPCache : 1];
  if (p_cached) {
    for (int i = tid; i < nb; i += kThrowThreads) s_P[i] = P[b0 + i];
    if (RNG_MODE == 1)
      for (int i = tid; i < nb - 1; i += kThrowThreads) s_bin[i] = load_throw_bin(P, NW, XP, YP, a.sigl, a.sigh, b0 + i);
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
    uint32_t wide_end = (RNG_MODE == 1) ? 0u : bin_start + (uint32_t)max(NW[b], 0);

    if (RNG_MODE == 1) {
      // The lane throws its units (whole blocks of kThrowBlock electrons) one after the other.  Inside a
      // block the loop counter is wave-uniform (a scalar); what differs between lanes is where the bin or
      // the sigma changes, so each lane keeps ONE number -- `stop`, the in-block index of its next such
      // event (or of the end of its electrons) -- and the per-electron work is the draw, the deposit and one
      // compare.  The body of the event branch runs when some lane of the wave needs it (~1 in 6 iterations
      // at ~350 wide electrons per bin) and is two LDS reads.
      ThrowBin cur;
      cur.x = cur.y = -1e30f; cur.ox = cur.oy = 0; cur.ch = cur.cl = 0.f; cur.bin_end = bin_start; cur.wide_end = bin_start;
      int bi = b - 1;                       // bin index of `cur`: the first event loads bin b
      for (uint64_t u = ub; u < ue; ++u) {
        const uint32_t e0 = (uint32_t)(u * kThrowBlock);
        const uint32_t limit = min((uint32_t)kThrowBlock, E - e0);     // electrons of this block
        SeededStream rng(a.seed, STAGE_THROW, (uint32_t)u, (uint32_t)k + a.subsample0, a.exposure);
        uint32_t stop = 0;                  // force the event branch at j = 0
        float c = 0.f, x = -1e30f, y = -1e30f;
        uint32_t oxl = 0u, oyl = 0u;        // the bin's pixel relative to the tile's corner
        for (uint32_t j = 0; j < kThrowBlock; ++j) {
          if (j >= stop) {
            const uint32_t e = e0 + j;
            if (j >= limit) {               // past the last electron of the sub-sample: draw on, deposit nothing
              x = y = -1e30f; c = 0.f; stop = 0xFFFFFFFFu;
            } else {
              while (e >= cur.bin_end && bi + 1 <= b1) {   // next populated bin
                ++bi;
                cur = p_cached ? s_bin[bi - b0] : load_throw_bin(P, NW, XP, YP, a.sigl, a.sigh, bi);
              }
              // the first n_wide electrons of a bin take the wide gaussian (pyparallel_menu.c:89-98)
              const bool wide = e < cur.wide_end;
              c = wide ? cur.ch : cur.cl;
              x = cur.x; y = cur.y; oxl = (uint32_t)cur.ox - (uint32_t)tx0; oyl = (uint32_t)cur.oy - (uint32_t)ty0;
              // a sigma that is not finite (or whose square is not): the reference's (int) of a non-finite position
              // keeps none of these electrons (:91-93) -- settled here, once per segment, so that the sums below are finite
              if (!(c > -3e38f)) { x = y = -1e30f; c = 0.f; }
              const uint32_t seg_end = wide ? cur.wide_end : cur.bin_end;
              stop = min(seg_end - e0, limit);
              if (e >= cur.bin_end) { x = y = -1e30f; c = 0.f; stop = 0xFFFFFFFFu; }   // inconsistent prefix: nothing to throw
            }
          }
          uint32_t wa, wb;
          rng.next2(wa, wb);
          // Box-Muller with the sigma folded in: R sigma = sqrt(-2 ln(ub)) sigma = sqrt((-2 ln2 sigma^2) log2(ub));
          // sin / cos take revolutions, and any window of length 1 will do: [1, 2) straight from the bits
          const float rev = rev12(wa);
          const float Rs = __builtin_amdgcn_sqrtf(c * __builtin_amdgcn_logf(u01f(wb)));
          // the bin's pixel + floor(offset + the bin's fraction of it) (:91-92; common.h, bin_local), counted from the
          // tile's corner (unsigned: a dead lane's -1e30 saturates the conversion and wraps to a cell off every frame)
          uint32_t lx, ly;
          asm("v_cvt_flr_i32_f32 %0, %2\n\tv_cvt_flr_i32_f32 %1, %3\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5"
              : "=&v"(lx), "=&v"(ly)
              : "v"(fmaf(__builtin_amdgcn_cosf(rev), Rs, x)), "v"(fmaf(__builtin_amdgcn_sinf(rev), Rs, y)), "v"(oxl), "v"(oyl));
          // the tile lies inside [1, N) x [1, N), so this one test implies the
          // reference's 0 < pos < n bounds (:93) on the fast path
          if (lx < (uint32_t)tw && ly < (uint32_t)th) {
            atomicAdd(&tile[__umul24(ly, tw) + lx], 1);   // tile area < 2^14: 24-bit multiply-add
          } else {
            const int xi = (int)(lx + (uint32_t)tx0), yi = (int)(ly + (uint32_t)ty0);
            if (xi > 0 && xi < a.N && yi > 0 && yi < a.N) deposit_global<FLUSH>(a, si, xi, yi, 1);
          }
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
      // The fp64 Box-Muller below is what the reference computes (pyparallel_menu.c:57-61, 87-93) and ~400 fp64
      // instructions per electron.  Its outcome is two integers, and a float32 evaluation on the hardware's sin / cos /
      // log2 / sqrt units (~30 instructions) gives the same two integers unless the position falls within the float32
      // path's error of a pixel boundary: THEN, and only then, the fp64 code runs (one electron in ~5000; a wave takes
      // the branch for ~1 % of its iterations).  The error bound is measured, not estimated -- exhaustively over the
      // 2^31 values of a rand_r call (scripts/ubench/replay_fast_error.hip, profiles/r03/replay_fast_error.txt):
      // |cos32 - cos64|, |sin32 - sin64| <= 2.7e-7; |R32 - R64| <= 3.0e-6 for R >= 0.01 (<= 5.8e-7 for R >= 0.1) -- so
      // the offset R sigma (cos, sin) is good to sigma (3.0e-6 + 6.56 x 2.7e-7) = 4.8e-6 sigma plus three float32
      // roundings of numbers below 64 (1.2e-5): `band` below is twice that.  R < 0.01 (5e-5 of the electrons), a
      // zero rand_r, positions or sigmas out of the sane range: fp64.  The frames stay the reference's bit for bit
      // (tests/test_psf_gpu.py: the 13 golden frames and random inputs; tests/test_fullsize_oracle_gpu.py: 10^9
      // electrons against the compiled reference C, no accumulator differs).
      struct Fast { int ix, iy; float fx, fy, sl, sh; bool ok; };
      auto fast_of = [](double x_, double y_, double sl_, double sh_) {
        Fast f;
        f.ok = fabs(x_) < 1e6 && fabs(y_) < 1e6 && sl_ > 0. && sl_ < 1e3 && sh_ > 0. && sh_ < 1e3;
        const double flx = floor(x_), fly = floor(y_);
        f.ix = f.ok ? (int)flx : 0; f.iy = f.ok ? (int)fly : 0;
        f.fx = (float)(x_ - flx); f.fy = (float)(y_ - fly);
        f.sl = (float)sl_; f.sh = (float)sh_;
        return f;
      };
      Fast fb = fast_of(x, y, sl, sh);
      for (; e < e_end; ++e) {
        if (e >= bin_end) {
          do { ++b; bin_start = bin_end; bin_end = P[b + 1]; } while (bin_end <= e && b + 1 < W);
          wide_end = bin_start + (uint32_t)max(NW[b], 0);
          x = XP[b]; y = YP[b]; sl = a.sigl[b]; sh = a.sigh[b];
          fb = fast_of(x, y, sl, sh);
        }
        while (e >= part_end && part + 1 < T) {   // next emulated thread: fresh stream
          ++part;
          part_end = (part == T - 1) ? E : part_start_of(part + 1);
          lcg = (uint32_t)(25234 + 17 * part + si.replay_seed);
        }
        const int k1 = rand_r_step(lcg), k2 = rand_r_step(lcg);
        const bool wide = e < wide_end;
        int xi, yi;
        {
          const float sg = wide ? fb.sh : fb.sl;
          const float rev = (float)k1 * 4.656612873077393e-10f;          // k1 2^-31 revolutions
          const float kf = (float)k2;
          // log2(k2 2^-31) = (e - 31) + log2 m, (float)k2 = m 2^e, m in [0.5, 1)
          const float t = (float)(__builtin_amdgcn_frexp_expf(kf) - 31) + __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(kf));
          const float r2 = t * -1.3862943611198906f;                     // R^2 = -2 ln u
          const float Rs = __builtin_amdgcn_sqrtf(r2) * sg;
          const float sx = fmaf(__builtin_amdgcn_cosf(rev), Rs, fb.fx);  // position - floor(bin position)
          const float sy = fmaf(__builtin_amdgcn_sinf(rev), Rs, fb.fy);
          const float flx = floorf(sx), fly = floorf(sy);
          const float band = fmaf(sg, 1e-5f, 2.5e-5f) + (fabsf(sx) + fabsf(sy)) * 2.4e-7f;
          const float gx = sx - flx, gy = sy - fly;
          const bool sure = fb.ok && r2 >= 1e-4f && gx > band && gx < 1.f - band && gy > band && gy < 1.f - band;
          xi = fb.ix + (int)flx; yi = fb.iy + (int)fly;
          if (!sure) {
            // pyparallel_menu.c:57-61
            const double theta = 2. * kPi * k1 / ((double)2147483647);
            const double R = sqrt(-2. * log(k2 / ((double)2147483647)));
            const double zx = R * cos(theta);
            const double zy = R * sin(theta);
            const double sig = wide ? sh : sl;
            const double px = zx * sig + x, py = zy * sig + y;
            // (int) of a non-finite / out-of-range double: reject (x86 gives INT_MIN)
            const bool okx = (px > -2147483649.0 && px < 2147483648.0);
            const bool oky = (py > -2147483649.0 && py < 2147483648.0);
            xi = okx ? (int)px : -1; yi = oky ? (int)py : -1;
          }
        }
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

}  // namespace wayne
