// k_ramp: per-read and post-ramp stages (A13-A15); the cosmic-ray hits of A13 are added by k_prep_sub (k_prep.h)
#pragma once
#include "common.h"

namespace wayne {

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
  // the same per-read numbers laid out for wave-uniform (scalar) loads in the production variant's read loop
  float bg[16];              // (float)(sky_ct_s * read_dt[r]): bg_count of read r (:489-491)
  int tab0[16];              // sky_tab0[r]
  // Which accumulators of read interval r can be non-zero: those inside box[r] = {x0, x1, y0, y1} (bordered
  // coordinates, half open; the host's bound on where the thrower's electrons of that interval can land) and those of
  // a 64-pixel segment whose bit r is set in `seg` (cosmic-ray hits).  A wave loads its 64 accumulators of a read
  // only when one of the two says so -- ~95 % of a frame never leaves zero.  use_box = 0: load everything.
  int use_box;
  int box[16][4];
  uint32_t* seg;             // [ceil(S*S / 64)] read and cleared
};

// (kSkyAlias, kSkyPiece, kMaxReads: plan_consts.h)

#ifndef WAYNE_RAMP_THREADS
#define WAYNE_RAMP_THREADS 1024
#endif
constexpr int kRampThreads = WAYNE_RAMP_THREADS;

// Box-Muller pair from two words.  EXACT mirrors the oracle's libm formula;
// FAST uses the hardware units (sin / cos take revolutions, log is log2).
template <bool FAST>
__device__ __forceinline__ void bm_pair(uint32_t w0, uint32_t w1, float& z0, float& z1) {
  // angle: 23 bits of w0 as a float in [1, 2) -- revolutions, one instruction -- radius from u01f(w1)
  const float rev = rev12(w0), ub = u01f(w1);
  if (FAST) {
    const float Rr = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(ub));
    z0 = Rr * __builtin_amdgcn_cosf(rev);
    z1 = Rr * __builtin_amdgcn_sinf(rev);
  } else {
    const float Rr = sqrtf(-2.0f * logf(ub));
    const float ang = 6.283185307179586f * (rev - 1.0f);
    z0 = Rr * cosf(ang);
    z1 = Rr * sinf(ang);
  }
}

// The state that lets the Newton solve of read r start next to its root (see nonlinear_response)
struct NlState {
  float v_prev;     // the previous read's incoming value
  float gap;        // v_prev - u_prev: what the non-linearity took off it
  float bend;       // 1 - 1/f'(u_prev): how that gap grows per DN
};

__device__ __forceinline__ double nonlinear_response(double px, float c1, float c2, float c3, float c4, NlState& st) {
  // WFC3_IR.apply_non_linearity (detector.py:335-348): Newton-Raphson on
  // u (1 + c1 + u (c2 + u (c3 + c4 u))) = px, until |du| < 1e-3.
  // The reference starts every read from u0 = px and iterates the whole frame until its slowest pixel
  // converges; here each pixel stops on its own criterion, and starts from the previous read's root
  // moved along the previous read's slope, u0 = px - gap - (px - v_prev) bend: the value of a pixel
  // grows by a few DN to a few hundred DN per read, so u0 is within ~c2 dv^2 of the root and the first
  // step is already below 1e-3 (one evaluation instead of two or three; read 1 starts from px).  The
  // iteration converges quadratically, so whatever the start the result is the root to < 1e-9 DN.
  // (1 + c1), 2*c2, 3*c3, 4*c4 are float32 in the reference because the coefficient planes are.  The
  // residual is evaluated in fp64; the reciprocal of the derivative (within 2e-7 of 1 + small) in fp32,
  // which changes an iterate by < 1e-7 of its step.
  const double k1 = (double)(1.0f + c1);
  const double k2 = (double)c2, k3 = (double)c3, k4 = (double)c4;
  const float d1 = 1.0f + c1, d2 = 2.0f * c2, d3 = 3.0f * c3, d4 = 4.0f * c4;
  const float pxf = (float)px;
  double u0 = px - (double)fmaf(pxf - st.v_prev, st.bend, st.gap), u1 = u0;
  float rinv = 1.0f;
  for (int it = 0; it < 10000; ++it) {
    const double f = fma(u0, fma(u0, fma(u0, fma(k4, u0, k3), k2), k1), -px);
    const float uf = (float)u0;
    const float fp_ = fmaf(uf, fmaf(uf, fmaf(uf, d4, d3), d2), d1);
    rinv = __builtin_amdgcn_rcpf(fp_);
    const double step = f * (double)rinv;
    u1 = u0 - step;
    if (fabs(step) < 1e-3) break;
    u0 = u1;
  }
  st.v_prev = pxf;
  st.gap = (float)(px - u1);
  st.bend = 1.0f - rinv;
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
  // shared part: N ~ Poisson(lam_level) from the first word of the pair, the remainder's first uniform from the second
  uint32_t w, wr;
  rng.next2(w, wr);
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
      float u = M::u01(wr);
      float pk = M::exp_(-piece);
      float j = 0.f;
      for (int it = 0; it < 512; ++it) {
        if (u <= pk) break;
        // (a term that no longer moves u cannot end the search either: the uniform fell into the rounding residue of
        // the pmf's float32 sum -- 3e-8 to 9e-8 of the draws, one pixel in an exposure or two.  Without this stop the
        // search walks on to its cap: ~500 spurious electrons in that pixel and, on the device, a wave that runs tens
        // of microseconds after every other wave of the launch has finished)
        const float un = u - pk;
        if (un == u) break;
        u = un;
        j = j + 1.f;
        pk = pk * M::div_(piece, j);
      }
      k = k + j;
      if (!PIECES) break;
      ld = ld - piece;
      if (!(ld > 0.f)) break;
      wr = rng.next();           // further pieces: one more word each
    }
  }
  return k;
}

// nonlinear_gap with the pixel value itself in float32 (the all-float32 production chain, see ramp_body): the same
// iteration, the same stop, the same state; returns u = px - g rounded once.
__device__ __forceinline__ float nonlinear_gap_f32(float px, float c1p, float c2, float c3, float c4, NlState& st) {
  const float d1 = 1.0f + c1p, d2 = 2.0f * c2, d3 = 3.0f * c3, d4 = 4.0f * c4;
  float g = fmaf(px - st.v_prev, st.bend, st.gap);
  float rinv = 1.0f, step = 0.f;
  auto newton = [&]() {
    const float u = px - g;
    const float h = fmaf(u, fmaf(u, fmaf(u, c4, c3), c2), c1p);
    const float fp_ = fmaf(u, fmaf(u, fmaf(u, d4, d3), d2), d1);
    rinv = __builtin_amdgcn_rcpf(fp_);
    step = fmaf(u, h, -g) * rinv;
    g = g + step;
  };
  // The first evaluation stands outside the loop: from the warm start it is the only one practically every pixel needs,
  // and as the first trip of a counted loop it paid for the loop's counter (a vector register, one v_subrev per trip)
  // and its exec-mask bookkeeping on every read.  Same iterates, same stop, same cap of 64 evaluations.
  newton();
  if (!(fabsf(step) < 1e-3f)) {
    for (int it = 1; it < 64; ++it) {
      newton();
      if (fabsf(step) < 1e-3f) break;
    }
  }
  st.v_prev = px;
  st.gap = g;
  st.bend = 1.0f - rinv;
  return px - g;
}

// The remainder's first four cumulative probabilities e^-m (1, 1 + m, 1 + m + m^2/2, 1 + ... + m^3/6) as 32-bit integer
// thresholds of the random WORD: u01f(w) > c  <=>  w > c 2^32 up to the float32 rounding of u01f (6e-8 of the draws
// decide differently from the float compare -- the hardware e^-m is itself only good to 1e-6).  They change only
// when the read interval does, so the read loop keeps them in registers: the search is then four integer compares
// feeding v_addc, with no conversion of the word.
struct SkyRem {
  float m;                   // the pixel's remainder mean (sky_px - level) * bg_count
  uint32_t t0, t1, t2, t3;
  float c3, term3;           // the float cdf and its last term, for the rare continuation beyond four
  // c 2^32 as a word, saturating: c >= 1 -> 0xFFFFFFFF, a threshold no word exceeds.  cdf >= 1 happens routinely (m = 0
  // gives exactly 1, the rounded partial sums of e^-m can exceed it), and a float -> uint32 conversion out of range is
  // undefined in C++ (poison in LLVM) whatever v_cvt_u32_f32 does with it: the saturation is spelled out, the
  // conversion only ever sees values below 2^32 (tests/test_extremes_gpu.py: pixels that sit exactly on their level)
  static __device__ __forceinline__ uint32_t thr(float cdf) {
    const float x = cdf * 4294967296.f;
    return (x < 4294967296.f) ? (uint32_t)x : 0xFFFFFFFFu;
  }
  __device__ __forceinline__ void set(float m_) {
    m = m_;
    float em;
    asm volatile("v_exp_f32 %0, %1" : "=v"(em) : "v"(-1.4426950408889634f * m_));   // (volatile: not to be speculated into every read)
    float t = em, cdf = t;
    t0 = thr(cdf);  t = t * m_;                  cdf += t;
    t1 = thr(cdf);  t = t * (m_ * 0.5f);         cdf += t;
    t2 = thr(cdf);  t = t * (m_ * 0.33333334f);  cdf += t;
    t3 = thr(cdf);
    c3 = cdf; term3 = t;
  }
};

template <class RNG>
__device__ __forceinline__ int sky_draw_count_int(const uint32_t* tab, const SkyRem& sr, RNG& rng) {
  uint32_t w, wr;
  rng.next2(w, wr);
  const uint32_t idx = w >> 24;
  const uint32_t e = tab[idx];
  int k = (int)(((w & 0xFFFFFFu) < (e & 0xFFFFFFu)) ? idx : (e >> 24));
  k += (wr > sr.t0) ? 1 : 0;
  k += (wr > sr.t1) ? 1 : 0;
  k += (wr > sr.t2) ? 1 : 0;
  if (wr > sr.t3) {
    k += 1;
    const float u = u01f(wr);
    float t = sr.term3, cdf = sr.c3;
#pragma nounroll
    for (int it = 4; it < 512; ++it) {
      t = t * FastMath::div_(sr.m, (float)it);
      const float cn = cdf + t;
#ifndef WAYNE_NEGCTL_SKY_RUNAWAY     // (negative-control builds of tests/test_extremes_gpu.py bring the defect of rounds 1-3 back)
      if (cn == cdf) break;          // the cdf has stopped growing below u (see sky_draw): stop, not 500 more rounds
#endif
      cdf = cn;
      if (!(u > cdf)) break;
      k = k + 1;
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
        uint32_t w1, w2;
        rng.next2(w1, w2);
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
// NOISE: the optional gaussian noise stage (noise_mean / noise_std; off in every shipped configuration) is
// compiled in or out: its stream would otherwise hold four registers of a kernel that lives at the 64-VGPR limit
template <class OutT, bool FAST, int SKY, bool NOISE, bool ALLON>
__device__ __forceinline__ void ramp_body(const RampArgs& a) {
  constexpr bool ALIAS = SKY != 0;
  typedef typename std::conditional<FAST, FastMath, ExactMath<float> >::type M;
  static_assert(kSkyAlias <= kRampThreads, "the sky tables and the per-thread sky counts share one LDS array");
  // ALIAS: alias tables [table][entry]; else: sky counts [read][thread]
  __shared__ uint32_t s_tab[kMaxReads][ALIAS ? kSkyAlias : kRampThreads];
  __shared__ float s_c[kMaxReads + 1];
  __shared__ int s_tab0[kMaxReads + 1];   // first alias table of read r (a per-read index into the kernel arguments would be a global load)
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
  const bool do_noise = NOISE && (a.noise_mean != 0.) && (a.noise_std != 0.);   // `if noise_mean and noise_std` (:477)
  const bool do_sky = a.sky_ct_s > 0. && a.sky;

  // Buffer addressing: descriptor (scalar) + the lane's 32-bit byte offset (vector, loop constant) + the plane's
  // byte offset (scalar, advances per read) -- no 64-bit vector address arithmetic per access, which the
  // flat-pointer form spent five v_lshl_add_u64 a read on.  (Planes are < 2^31 bytes: R * S * S * 8 <= 142 MB.)
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const int n_acc = (int)((size_t)a.R * SS * sizeof(long long)), n_f32 = (int)((size_t)a.R * SS * sizeof(float));
  const int n_out = (int)((size_t)(a.R + 1) * SS * sizeof(OutT));
  const __amdgpu_buffer_rsrc_t rs_acc = __builtin_amdgcn_make_buffer_rsrc((void*)a.acc, 0, n_acc, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ds = __builtin_amdgcn_make_buffer_rsrc((void*)a.dark_sci, 0, a.dark_sci ? n_f32 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_de = __builtin_amdgcn_make_buffer_rsrc((void*)a.dark_err, 0, a.dark_err ? n_f32 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, n_out, 0x00020000);
  const uint32_t off8 = (uint32_t)p * 8u, off4 = (uint32_t)p * 4u, offo = (uint32_t)p * (uint32_t)sizeof(OutT);
  const uint32_t acc_plane = (uint32_t)(SS * sizeof(long long)), f32_plane = (uint32_t)(SS * sizeof(float));
  const uint32_t out_plane = (uint32_t)(SS * sizeof(OutT));
  constexpr int kNT = 2;                                   // cache policy: non-temporal (streamed once)
  auto ld_acc = [&](int r) -> long long {
    const v2u w = __builtin_amdgcn_raw_buffer_load_b64(rs_acc, off8, (uint32_t)r * acc_plane, 0);
    return (long long)(((unsigned long long)w.y << 32) | w.x);
  };
  auto ld_f32 = [&](const __amdgpu_buffer_rsrc_t& rs, int r) -> float {
#ifdef WAYNE_TIMING_RAMP_NO_DARK
    // TIMING BUILD (wrong frames; HISTORY.md section 9, "two exposures per launch"): the dark planes for free -- what the
    // second exposure of a pair that shared its partner's dark loads would cost
    return 0.02f + 1e-9f * (float)(r + (int)off4);
#endif
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off4, (uint32_t)r * f32_plane, kNT));
  };
  auto st_out = [&](int plane, OutT v) {
    if (sizeof(OutT) == 4) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)v), rs_out, offo, (uint32_t)plane * out_plane, kNT);
    } else {
      const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
      const v2u w = {(unsigned)b, (unsigned)(b >> 32)};
      __builtin_amdgcn_raw_buffer_store_b64(w, rs_out, offo, (uint32_t)plane * out_plane, kNT);
    }
  };

  // ---- prologue.  Every load whose address is known at entry is ISSUED first -- the once-per-pixel planes, this
  // wave's per-read numbers (lane l fetches those of read l), its share of the alias tables -- then the stream is
  // seeded (a Philox block: ~100 vector instructions that need none of them), and only then are the values used.  In
  // the order the stages are written below -- load, use, load, use -- a wave paid eight memory latencies one after the
  // other before its first read, twice per launch (two rounds of workgroups): 17 us of a 54 us kernel
  // (scripts/ramp_vs_reads.py).
  const int lane_r = min(tid & 63, kMaxReads);
  const int p0 = __builtin_amdgcn_readfirstlane(p_raw) & ~63;
#ifdef WAYNE_TIMING_RAMP_NO_ONCE
  // TIMING BUILD (wrong frames; HISTORY.md section 9): the once-per-pixel planes for free -- what a second exposure in
  // the same launch would save on them
  const float t_sky = (interior && do_sky) ? 1.0f + 1e-4f * (float)(p & 255) : 0.f;
  const float t_pfl = 1.0f + 1e-5f * (float)(p & 127);
  float c1 = 0, c2 = 7e-7f + 1e-12f * (float)(p & 63), c3 = 0, c4 = 0;
#else
  const float t_sky = (interior && do_sky) ? a.sky[p] : 0.f;
  const float t_pfl = (interior && gainvar) ? a.pfl[p] : 1.0f;
  float c1 = 0, c2 = 0, c3 = 0, c4 = 0;
  if (do_lin && valid) { c1 = a.lin[0][p]; c2 = a.lin[1][p]; c3 = a.lin[2][p]; c4 = a.lin[3][p]; }
#endif
  const double t_zero = (valid && a.zero_read && (a.flags & (1u << 7))) ? a.zero_read[p] : 0.;
  const int v_bg = __float_as_int(a.bg[lane_r]), v_tab0 = a.tab0[lane_r];
  const int v_lvl = __float_as_int(a.sky_level[lane_r]);
  int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
  uint32_t t_seg = 0u;
  if (a.use_box) {
    bx0 = a.box[lane_r][0]; bx1 = a.box[lane_r][1]; by0 = a.box[lane_r][2]; by1 = a.box[lane_r][3];
    if (p0 < S * S) t_seg = a.seg[p0 >> 6];
  }
  constexpr int kTabWords = kMaxReads * kSkyAlias, kTabPer = (kTabWords + kRampThreads - 1) / kRampThreads;
  uint32_t t_tab[ALIAS ? kTabPer : 1];
  if (ALIAS && do_sky) {
#pragma unroll
    for (int i = 0; i < kTabPer; ++i) {
      const int j = tid + i * kRampThreads;
      t_tab[i] = j < kTabWords ? a.sky_alias[j] : 0u;
    }
  }

  // per-pixel streams, seeded only when a stage that reads them is on (one Philox block each).  ONE stream (STAGE_READ)
  // serves the table-driven sky draw and the two normals of every read, in the order the read loop takes them: pair 0 =
  // the zero read's normals, then per read interval r the words of its sky draw (a pair; none where the pixel has no
  // sky) followed by the pair of read r + 1's normals.  A second Philox block per pixel for the sky alone was 6 % of the
  // kernel's vector instructions.  (The direct sampler, SKY = 0, draws a data-dependent number of words per read and
  // keeps its own stream, STAGE_SKY: sky_counts.)
  // (the first read's dark planes too; reference pixels load theirs as well -- the production loop runs them through
  // every stage and zeroes them at the end)
  const bool ld_dark = do_dark && valid;
  float ds_next = ld_dark ? ld_f32(rs_ds, 0) : 0.f, de_next = ld_dark ? ld_f32(rs_de, 0) : 0.f;

  SeededStream rn, rg;
  if (rdn || do_dark || (ALIAS && do_sky)) rn = SeededStream(a.seed, STAGE_READ, (uint32_t)p, 0u, a.exposure);
  SeededStream& rs = rn;
  if (do_noise) rg = SeededStream(a.seed, STAGE_NOISE, (uint32_t)p, 0u, a.exposure);
  // (nothing loaded above is touched before this point: the compiler would otherwise sink a first use -- and its wait --
  // into the branch that issued the load)
  asm volatile("" : "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4));
  asm volatile("" : "+v"(ds_next), "+v"(de_next));

  if (tid < a.R) {
    s_c[tid] = __int_as_float(v_bg);                                       // bg_count of read tid (:489-491), lane = read
    s_tab0[tid] = v_tab0;
  }
  if (ALIAS && do_sky) {
#pragma unroll
    for (int i = 0; i < kTabPer; ++i) {
      const int j = tid + i * kRampThreads;
      if (j < kTabWords) (&s_tab[0][0])[j] = t_tab[i];
    }
  }
  const float skyv = t_sky;
  __syncthreads();
  if (!ALIAS) {
    if (do_sky) sky_counts<FAST>(a, (uint32_t)p, tid, interior && skyv > 0.f, skyv, s_c,
                                 (uint32_t (*)[kRampThreads])&s_tab[0][0]);
    __syncthreads();
  }
  if (!valid) return;
  // the pixel's sky level (constant over the reads): the highest level not above its sky value (lane l of the wave
  // holds level l: no memory access in the search)
  int sky_lvl = 0;
  float sky_base = __int_as_float(__builtin_amdgcn_readlane(v_lvl, 0));
  if (ALIAS && skyv > 0.f) {
    for (int l = 1; l < a.sky_levels; ++l) {
      const float lv = __int_as_float(__builtin_amdgcn_readlane(v_lvl, l));
      if (lv <= skyv) { sky_lvl += 1; sky_base = lv; }
    }
  }

  // Per-read numbers without a memory access in the read loop: lane l of every wave fetches those of read l once
  // (bg_count, first sky table, the box of the read's accumulators), tests the wave's 64 consecutive pixels against
  // box l, and the loop then takes read r's numbers with v_readlane and its "load the accumulators?" bit from a
  // ballot.  (Scalar loads of a.bg[r] / a.box[r] at the top of every iteration measured no gain over the LDS reads
  // they replaced: their latency sat in front of the iteration's first use.)
  const int wy0 = p0 / S, wy1 = min(p0 + 63, S * S - 1) / S, wx0 = p0 - wy0 * S;
  unsigned long long live_bits = ~0ull;
  if (a.use_box) {
    uint32_t cbits = 0u;
    if (p0 < S * S) {
      cbits = __builtin_amdgcn_readfirstlane(t_seg);
      if (cbits != 0u && (tid & 63) == 0) a.seg[p0 >> 6] = 0u;   // left clean for the next exposure, like the accumulators
    }
    const bool rows = wy0 < by1 && wy1 >= by0;
    const bool cols = (wy0 != wy1) || (wx0 < bx1 && wx0 + 64 > bx0);
    live_bits = __ballot((rows && cols) || ((cbits >> lane_r) & 1u) != 0u);
  }
  auto acc_live = [&](int r) -> bool { return ((live_bits >> r) & 1ull) != 0ull; };

  // zero read: (initial bias) -> clip -> reference pixels := 0 -> read noise
  // (exposure_generator.py:446-466, exposure.py:82-131, 61-68)
  double z = t_zero;
  if (clip) z = fmin(fmax(z, kMinCounts), kMaxCounts);
  if (!interior) z = 0.;
  {
    float zd, zr;
    uint32_t w0, w1;
    rn.next2(w0, w1);
    double v = z;
    if (rdn) { bm_pair<FAST>(w0, w1, zd, zr); v = v + kReadNoise * (double)zr; }
    st_out(0, (OutT)v);
  }

  // gain: 2.35 / pfl evaluated in float32 as numpy does for scalar / f32 array
  // (detector.py:203-204), or the constant (exposure_generator.py:507-511);
  // applied as a multiplication by its fp64 reciprocal
  const float g32 = 2.35f / t_pfl;                          // (float32 IEEE division, as numpy's)
  double inv_g = 1.0 / kGain;
  if (interior && gainvar && !(std::is_same<OutT, float>::value && FAST && SKY == 1 && !NOISE)) inv_g = 1.0 / (double)g32;

  // software-pipelined ramp: the planes of read r+1 are requested before the
  // (VALU-heavy) work on read r so that HBM latency hides behind it.  The dark planes and the reads are
  // streamed once: non-temporal loads / stores (kNT; measured: 0.076 -> 0.070 ms)
  long long q_next = (interior && acc_live(0)) ? ld_acc(0) : 0;
  double cum = 0.;
  NlState nl = {0.f, 0.f, 0.f};
  float sky_c = -1.f;
  SkyRem srem;
  srem.m = 0.f; srem.t0 = srem.t1 = srem.t2 = srem.t3 = 0xFFFFFFFFu; srem.c3 = 1.f; srem.term3 = 0.f;
  if (std::is_same<OutT, float>::value && FAST && SKY == 1 && !NOISE) {
    // PRODUCTION VARIANT (float32 reads, hardware math, alias-table sky, no gaussian-noise stage): the same stages,
    // streams and draws as the generic loop below with the arithmetic cut to what a float32 read can tell.
    //  * what a pixel has collected so far is kept EXACTLY, in integers: Q = the sum of its fixed-point accumulators
    //    (int64; touched only by the waves a read's electrons can reach, ~5 % of them) and ksum = the sum of its sky
    //    counts (int32).  One conversion per read takes it to float32 electrons -- a single rounding of the exact sum,
    //    where a float32 running sum would round fifteen times -- and one multiply to DN (the reference's
    //    sum_r (px_r / gain) is (sum_r px_r) / gain to 1e-16);
    //  * dark, non-linearity (Newton on the gap, nonlinear_gap_f32), clip, zero read and read noise follow in float32:
    //    five roundings in all, <= 3 ulp of the float32 read (tests/test_modes_gpu.py: 0.02 DN + 2e-7 against the
    //    float64 variant; tests/test_fullsize_oracle_gpu.py against the oracle);
    //  * the sky remainder is four integer compares of the random word against thresholds that change only with the
    //    read interval (SkyRem); the per-read numbers (bg_count, first table) come in scalar registers;
    //  * reference pixels run the same instructions as light-sensitive ones (their planes hold the border's fill values,
    //    their accumulators and their master sky are zero) and are set to zero by one select at the end: no exec-mask
    //    bracket around each stage;
    //  * ALLON (a kernel instantiation of its own, chosen by the host): every detector switch of the exposure is on --
    //    dark, non-linearity, clip, read noise: the rule -- so no stage is selected by a flag at run time.
    {
      const bool f_dark = ALLON || do_dark, f_lin = ALLON || do_lin, f_clip = ALLON || clip, f_rdn = ALLON || rdn;
      // DN per electron (reference pixels: nothing collected): the float32 reciprocal of the float32 gain
      const float inv_gf = !interior ? 0.f : (gainvar ? 1.0f / g32 : (float)(1.0 / kGain));
      const float c1p = (1.0f + c1) - 1.0f;
      const float zf = (float)z;
      const bool has_sky_px = skyv > 0.f;                  // (the pixel's own half of `skyv * bg > 0`: constant over the reads)
      uint32_t bg_prev = 0xFFFFFFFFu;                      // bits of the previous read's bg (a scalar, like bg)
      unsigned long long Q = 0ull;
      int ksum = 0;
      bool q_any = false;                                  // wave-uniform: has any read of this wave been live?
      SkyRem sr;
      sr.m = 0.f; sr.t0 = sr.t1 = sr.t2 = sr.t3 = 0xFFFFFFFFu; sr.c3 = 1.f; sr.term3 = 0.f;
#ifndef WAYNE_RAMP_PF
#define WAYNE_RAMP_PF 2
#endif
      // Register sets for the planes of a read (A, B[, C]): a step works on one while the loads of the read PF steps
      // ahead fill another, and the loop is written out in groups of PF + 1 steps so that no register is copied to
      // rotate them.
      constexpr int PF = WAYNE_RAMP_PF;
      long long qA = q_next, qB = 0, qC = 0;
      float dsA = ds_next, deA = de_next, dsB = 0.f, deB = 0.f, dsC = 0.f, deC = 0.f;
      if (PF == 2 && a.R > 1) {
        if (acc_live(1)) qB = ld_acc(1);
        if (f_dark) { dsB = ld_f32(rs_ds, 1); deB = ld_f32(rs_de, 1); }
      }
      auto step = [&](const int r, const long long q, const float ds, const float de, long long& q_nx, float& ds_nx,
                      float& de_nx) {
        const bool live = acc_live(r);                     // scalar
        if (r + PF < a.R) {
          if (acc_live(r + PF)) q_nx = ld_acc(r + PF);
          if (f_dark) { ds_nx = ld_f32(rs_ds, r + PF); de_nx = ld_f32(rs_de, r + PF); }
        }
        const float bg = __int_as_float(__builtin_amdgcn_readlane(v_bg, r));   // wave-uniform (see lane_r)
        const int tab = __builtin_amdgcn_readlane(v_tab0, r);
        const uint32_t bg_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(bg));
        if (bg_bits != bg_prev) {                          // a new read interval (a scalar branch): the pixel's
          bg_prev = bg_bits;                               // remainder mean and the thresholds of its search
          sr.set(fmaxf(skyv * bg - sky_base * bg, 0.f));
        }
        if (live) {                                        // (scalar branch)
          if (q != 0) __builtin_amdgcn_raw_buffer_store_b64(v2u{0u, 0u}, rs_acc, off8, (uint32_t)r * acc_plane, 0);
          Q += (unsigned long long)q;
          q_any = true;
        }
        if (has_sky_px && bg > 0.f) ksum += sky_draw_count_int(s_tab[tab + sky_lvl], sr, rn);   // (skyv = 0 off the sky; bg is the wave's)
        float zd = 0.f, zr = 0.f;
        uint32_t w0, w1;
        rn.next2(w0, w1);
        if (f_rdn || f_dark) bm_pair<true>(w0, w1, zd, zr);
        float e = (float)ksum;                             // electrons so far: sky + accumulators (hi 2^4 + lo 2^-28)
        if (q_any) {                                       // (scalar branch; the asm keeps it one: 95 % of the waves skip it)
          asm volatile("" ::: "memory");
          e = fmaf((float)(int)(Q >> 32), 16.0f, fmaf((float)(uint32_t)Q, 3.725290298461914e-09f, e));
        }
        float v = e * inv_gf;
        if (f_dark) v = v + fmaf(de, zd, ds);              // (de: already max(err, 1e-5), see wayne_ctx_set_calibration)
        if (f_lin) v = nonlinear_gap_f32(v, c1p, c2, c3, c4, nl);
        if (f_clip) v = __builtin_amdgcn_fmed3f(v, (float)kMinCounts, (float)kMaxCounts);
        if (!interior) v = 0.f;                            // reference pixels (exposure.py:122-131)
        // + zero read + read noise (exposure.py:94-104, detector.py:193-198)
        const float tail = f_rdn ? fmaf((float)kReadNoise, zr, zf) : zf;
        st_out(r + 1, (OutT)(v + tail));
      };
      int r = 0;
      if (PF == 1) {
        for (; r + 1 < a.R; r += 2) {
          step(r, qA, dsA, deA, qB, dsB, deB);
          step(r + 1, qB, dsB, deB, qA, dsA, deA);
        }
        if (r < a.R) step(r, qA, dsA, deA, qB, dsB, deB);
      } else {
        for (; r + 2 < a.R; r += 3) {
          step(r, qA, dsA, deA, qC, dsC, deC);
          step(r + 1, qB, dsB, deB, qA, dsA, deA);
          step(r + 2, qC, dsC, deC, qB, dsB, deB);
        }
        if (r < a.R) step(r, qA, dsA, deA, qC, dsC, deC);
        if (r + 1 < a.R) step(r + 1, qB, dsB, deB, qA, dsA, deA);
      }
    }
    return;
  }
  for (int r = 0; r < a.R; ++r) {
    const long long q = q_next;
    const float ds = ds_next, de = de_next;
    if (r + 1 < a.R) {
      q_next = 0; if (interior && acc_live(r + 1)) q_next = ld_acc(r + 1);
      if (ld_dark) { ds_next = ld_f32(rs_ds, r + 1); de_next = ld_f32(rs_de, r + 1); }
    }
    double px = 0.;
    uint32_t g0 = 0u, g1 = 0u;
    if (do_noise) rg.next2(g0, g1);
    if (interior) {
      // leave the accumulator clean for the next exposure -- 90 % of a frame never left zero, and not re-zeroing
      // those saves a quarter of the kernel's HBM traffic
      if (q != 0) __builtin_amdgcn_raw_buffer_store_b64(v2u{0u, 0u}, rs_acc, off8, (uint32_t)r * acc_plane, 0);
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
        if (ALIAS && FAST && SKY == 1) {
          if (s_c[r] != sky_c) {                         // a new read interval: the pixel's remainder mean and thresholds
            sky_c = s_c[r];
            srem.set(fmaxf(lam - sky_base * sky_c, 0.f));
          }
          if (lam > 0.f) px = px + (double)sky_draw_count_int(s_tab[s_tab0[r] + sky_lvl], srem, rs);
        } else if (ALIAS) {
          if (lam > 0.f) px = px + (double)sky_draw<M, SKY == 2>(s_tab[s_tab0[r] + sky_lvl], sky_base * s_c[r], lam, rs);
        } else {
          px = px + (double)s_tab[r][tid];
        }
      }
      px = px * inv_g;               // electrons -> DN (:507-511)
    }
    cum = cum + px;                  // cumulative_pixel_array += pixel_array_full (:378)
    double v = cum;
    float zd = 0.f, zr = 0.f;
    uint32_t w0, w1;
    rn.next2(w0, w1);
    if (rdn || ld_dark) bm_pair<FAST>(w0, w1, zd, zr);
    if (interior) {
      if (do_dark) {                 // detector.py:185-191
        const double err = (de > 0.f) ? (double)de : (double)0.00001f;
        v = v + ((double)ds + err * (double)zd);
      }
      if (do_lin) v = nonlinear_response(v, c1, c2, c3, c4, nl);
      if (clip) v = fmin(fmax(v, kMinCounts), kMaxCounts);
    } else {
      v = 0.;                        // reset_reference_pixels (exposure.py:122-131)
    }
    v = v + z;                       // add_zero_read (exposure.py:94-104)
    if (rdn) v = v + kReadNoise * (double)zr;   // add_read_noise (detector.py:193-198)
    st_out(r + 1, (OutT)v);
  }
}

// The production-math variants with a table-driven sky fit 64 registers -- 8 waves per SIMD, which is what hides the
// plane loads behind other waves' arithmetic (at 66 registers and 7 waves the kernel measured 7 % slower) -- and are
// pinned there.  The exact-math, direct-sky (SKY = 0) and gaussian-noise variants would spill under that pin
// (24-64 bytes of scratch each); they are parity / fallback paths and take the registers they need.
template <class OutT, bool FAST, int SKY, bool NOISE, bool ALLON = false>
__global__ __launch_bounds__(kRampThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_ramp(RampArgs a) {
  ramp_body<OutT, FAST, SKY, NOISE, ALLON>(a);
}
template <class OutT, bool FAST, int SKY, bool NOISE>
__global__ __launch_bounds__(kRampThreads) void k_ramp_wide(RampArgs a) {
  ramp_body<OutT, FAST, SKY, NOISE, false>(a);
}

}  // namespace wayne
