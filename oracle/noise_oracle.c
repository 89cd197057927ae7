/*
 * oracle/noise_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU statement of the Philox-keyed Poisson / normal / uniform-integer draws
 * that stand in for numpy's legacy generator on the MI355X path
 * (reference call sites: exposure_generator.py:327-329,495,626,725;
 *  detector.py:191,198; cosmic_rays.py:80-81,127,134).
 *
 * Parity status: the reference's MT19937 stream cannot be reproduced by a
 * sharded, counter-based generator, so these draws are "parity unpinned"
 * against the reference; they are pinned instead by
 *   (a) Random123's known-answer vectors for Philox4x32-10,
 *   (b) distribution tests against scipy.stats (tests/test_samplers.py),
 *   (c) agreement with the device under the same counters.
 * The algorithms are the published ones numpy's legacy poisson uses:
 * Knuth's product of uniforms for lam < 10 and Hoermann's PTRS transformed
 * rejection (Insurance: Mathematics and Economics 12, 1993) for lam >= 10;
 * the sky background of ordinary exposures is drawn as Poisson(level) from a
 * Walker / Vose alias table + Poisson(remainder) by sequential search.
 * Written separately from wayne_amd/csrc/samplers.h on purpose: two
 * statements of one specification check each other.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void wayne_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t wayne_oracle_xo_next(uint32_t state[4]);
void wayne_oracle_xo_next2(uint32_t state[4], uint32_t out[2]);
float wayne_oracle_rev12(uint32_t x);

/* A word source: either consecutive Philox blocks (counter word 1 = block
 * index) or a Philox-seeded xoshiro128+ state (`xo` != NULL). */
typedef struct {
  uint32_t ctr[4];
  uint32_t key[2];
  uint32_t buf[4];
  int have;
  uint32_t *xo;
} wo_stream;

static void wo_stream_init(wo_stream *s, uint32_t seed, uint32_t stage, uint32_t c0,
                           uint32_t c2, uint32_t c3) {
  s->ctr[0] = c0; s->ctr[1] = 0; s->ctr[2] = c2; s->ctr[3] = c3;
  s->key[0] = seed; s->key[1] = stage;
  s->have = 0;
  s->xo = 0;
}

static uint32_t wo_next(wo_stream *s) {
  if (s->xo) return wayne_oracle_xo_next(s->xo);
  if (s->have == 0) {
    wayne_oracle_philox4x32(s->ctr, s->key, s->buf);
    s->ctr[1] += 1;
    s->have = 4;
  }
  uint32_t v = s->buf[4 - s->have];
  s->have -= 1;
  return v;
}

/* A PAIR of words (the U, V of a rejection trial; angle and radius of a normal pair): one state transition
 * of a seeded stream (device: SeededStream::next2), two consecutive words of a Philox stream. */
static void wo_next2(wo_stream *s, uint32_t *a, uint32_t *b) {
  if (s->xo) {
    uint32_t w[2];
    wayne_oracle_xo_next2(s->xo, w);
    *a = w[0]; *b = w[1];
    return;
  }
  *a = wo_next(s);
  *b = wo_next(s);
}

static float wo_u01f(uint32_t x) { return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f); }
static double wo_u01d(uint32_t x) { return fma((double)x, 2.3283064365386963e-10, 1.1641532182693481e-10); }

/* ---- ln Gamma via shift + Stirling, double and float ------------------- */
static double wo_loggam_d(double x) {
  double prod = 1.0;
  for (int i = 0; i < 6 && x < 7.0; ++i) { prod = prod * x; x = x + 1.0; }
  const double xi = 1.0 / x, x2 = xi * xi;
  double s = -691.0 / 360360.0;
  s = s * x2 + 1.0 / 1188.0;
  s = s * x2 + -1.0 / 1680.0;
  s = s * x2 + 1.0 / 1260.0;
  s = s * x2 + -1.0 / 360.0;
  s = s * x2 + 1.0 / 12.0;
  s = s * xi;
  return (x - 0.5) * log(x) - x + 0.91893853320467274178 + s - log(prod);
}
static float wo_loggam_f(float x) {
  float prod = 1.0f;
  for (int i = 0; i < 6 && x < 7.0f; ++i) { prod = prod * x; x = x + 1.0f; }
  const float xi = 1.0f / x, x2 = xi * xi;
  float s = (float)(-691.0 / 360360.0);
  s = s * x2 + (float)(1.0 / 1188.0);
  s = s * x2 + (float)(-1.0 / 1680.0);
  s = s * x2 + (float)(1.0 / 1260.0);
  s = s * x2 + (float)(-1.0 / 360.0);
  s = s * x2 + (float)(1.0 / 12.0);
  s = s * xi;
  return (x - 0.5f) * logf(x) - x + (float)0.91893853320467274178 + s - logf(prod);
}

/* ---- Poisson ------------------------------------------------------------ */
static double wo_poisson_d(double lam, wo_stream *rng) {
  if (!(lam > 0.0)) return 0.0;
  if (lam < 10.0) {
    const double enlam = exp(-lam);
    double k = 0.0, prod = 1.0;
    for (int it = 0; it < 4096; ++it) {
      prod = prod * wo_u01d(wo_next(rng));
      if (prod > enlam) k = k + 1.0; else break;
    }
    return k;
  }
  const double slam = sqrt(lam), loglam = log(lam);
  const double b = 0.931 + 2.53 * slam;
  const double a = -0.059 + 0.02483 * b;
  const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
  const double vr = 0.9277 - 3.6224 / (b - 2.0);
  for (int it = 0; it < 256; ++it) {
    uint32_t wu, wv;
    wo_next2(rng, &wu, &wv);
    const double U = wo_u01d(wu) - 0.5;
    const double V = wo_u01d(wv);
    const double us = 0.5 - fabs(U);
    const double k = floor((2.0 * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return k;
    if (k < 0.0 || (us < 0.013 && V > us)) continue;
    const double lhs = log(V) + log(invalpha) - log(a / (us * us) + b);
    const double rhs = -lam + k * loglam - wo_loggam_d(k + 1.0);
    if (lhs <= rhs) return k;
  }
  return floor(lam + 0.5);
}

static float wo_poisson_f(float lam, wo_stream *rng) {
  if (!(lam > 0.0f)) return 0.0f;
  if (lam < 10.0f) {
    const float enlam = expf(-lam);
    float k = 0.0f, prod = 1.0f;
    for (int it = 0; it < 4096; ++it) {
      prod = prod * wo_u01f(wo_next(rng));
      if (prod > enlam) k = k + 1.0f; else break;
    }
    return k;
  }
  const float slam = sqrtf(lam), loglam = logf(lam);
  const float b = 0.931f + 2.53f * slam;
  const float a = -0.059f + 0.02483f * b;
  const float invalpha = 1.1239f + 1.1328f / (b - 3.4f);
  const float vr = 0.9277f - 3.6224f / (b - 2.0f);
  for (int it = 0; it < 256; ++it) {
    uint32_t wu, wv;
    wo_next2(rng, &wu, &wv);
    const float U = wo_u01f(wu) - 0.5f;
    const float V = wo_u01f(wv);
    const float us = 0.5f - fabsf(U);
    const float k = floorf((2.0f * a / us + b) * U + lam + 0.43f);
    if (us >= 0.07f && V <= vr) return k;
    if (k < 0.0f || (us < 0.013f && V > us)) continue;
    const float lhs = logf(V) + logf(invalpha) - logf(a / (us * us) + b);
    const float rhs = -lam + k * loglam - wo_loggam_f(k + 1.0f);
    if (lhs <= rhs) return k;
  }
  return floorf(lam + 0.5f);
}

/* The stellar counts of a bin (stage COUNTS, exposure_generator.py:626): PTRS as above from a mean of 10; below it
 * ONE uniform and a sequential search of the cdf from 0 (inversion) instead of the product of uniforms -- a finely
 * sampled scan draws millions of such counts per exposure, and the product of uniforms takes lam + 1 words (a
 * second and a third Philox block for a fifth and a ninth word) where inversion takes one.  The distribution is
 * Poisson(lam) either way; the draw is this project's own (the reference's comes from numpy's global stream). */
static double wo_poisson_counts(double lam, wo_stream *rng) {
  if (!(lam > 0.0)) return 0.0;
  if (!(lam < 10.0)) return wo_poisson_d(lam, rng);
  const double u = wo_u01d(wo_next(rng));
  double p = exp(-lam), c = p, k = 0.0;
  for (int it = 1; it < 256; ++it) {
    if (u <= c) break;
    p = p * lam / (double)it;
    c = c + p;
    k = (double)it;
  }
  return k;
}

void wayne_oracle_poisson_counts_f64(const double *lam, int64_t n, uint32_t seed, uint32_t stage,
                                     uint32_t c0_base, uint32_t c2, uint32_t c3, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    wo_stream s;
    wo_stream_init(&s, seed, stage, c0_base + (uint32_t)i, c2, c3);
    out[i] = wo_poisson_counts(lam[i], &s);
  }
}

/* Vector entry points used by oracle/wayne_oracle.py.  Element i draws from
 * the stream (c0[i] or c0_base + i, c2, c3) of stage `stage`. */
void wayne_oracle_poisson_f64(const double *lam, int64_t n, uint32_t seed, uint32_t stage,
                              uint32_t c0_base, uint32_t c2, uint32_t c3, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    wo_stream s;
    wo_stream_init(&s, seed, stage, c0_base + (uint32_t)i, c2, c3);
    out[i] = wo_poisson_d(lam[i], &s);
  }
}

/* Seeded streams (wayne_amd/csrc/philox.h): the 128-bit xoshiro128+ state of
 * element idx[i] is Philox block (idx[i], 0, 0, exposure) of `stage`. */
void wayne_oracle_seed_streams(const uint32_t *idx, int64_t n, uint32_t seed, uint32_t stage,
                               uint32_t exposure, uint32_t *state /* n*4 */) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t ctr[4] = {idx[i], 0u, 0u, exposure};
    const uint32_t key[2] = {seed, stage};
    wayne_oracle_philox4x32(ctr, key, state + 4 * i);
  }
}

/* One sky Poisson draw per pixel from its STAGE_SKY stream, advancing the
 * state (the ramp kernel draws read 0, 1, ... in this order): product of
 * uniforms below lam = 10, float32 PTRS below 256, float64 PTRS above. */
void wayne_oracle_poisson_sky_step(const float *lam, int64_t n, uint32_t *state, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    wo_stream s;
    s.xo = state + 4 * i;
    s.have = 0;
    out[i] = (lam[i] < 256.0f) ? (double)wo_poisson_f(lam[i], &s) : wo_poisson_d((double)lam[i], &s);
  }
}

/* ---- sky background through shared alias tables -------------------------
 * (device: wayne_amd/csrc/k_ramp.h sky_draw; reference call site
 *  exposure_generator.py:488-495, pixel += poisson(master_sky * bg_count)).
 * Poisson(sky_px * bg) = Poisson(level * bg) + Poisson((sky_px - level) * bg):
 * the first term from a Walker alias table shared by all pixels of one sky
 * level (levels = quantiles of the master sky), the second by sequential
 * search from 0.
 *
 * Table of Poisson(lam) over 0..255 (pmf by recurrence from the mode, then
 * normalised) by Vose's construction: columns are
 * visited from a stack of "small" (scaled probability < 1) and "large"
 * entries, both filled in ascending order and popped from the top; entry =
 * alias << 24 | round(prob * 2^24) capped at 2^24 - 1. */
void wayne_oracle_sky_alias_table(double lam, uint32_t *out /* 256 */) {
  enum { NT = 256 };
  double scaled[NT], keep[NT];
  int other[NT], lo_stack[NT], hi_stack[NT], n_lo = 0, n_hi = 0;
  double total = 0.0;
  if (lam > 0.0) {
    /* pmf by recurrence away from the mode: p(k+1) = p(k) lam / (k+1), p(k-1) = p(k) k / lam */
    int mode = (int)lam;
    if (mode > NT - 1) mode = NT - 1;
    scaled[mode] = exp(-lam + mode * log(lam) - lgamma(mode + 1.0));
    for (int k = mode + 1; k < NT; ++k) scaled[k] = scaled[k - 1] * lam / (double)k;
    for (int k = mode - 1; k >= 0; --k) scaled[k] = scaled[k + 1] * (double)(k + 1) / lam;
  } else {
    for (int k = 0; k < NT; ++k) scaled[k] = (k == 0) ? 1.0 : 0.0;
  }
  for (int k = 0; k < NT; ++k) total += scaled[k];
  for (int k = 0; k < NT; ++k) {
    scaled[k] = scaled[k] / total * NT;
    keep[k] = 1.0;
    other[k] = k;
    if (scaled[k] < 1.0) lo_stack[n_lo++] = k; else hi_stack[n_hi++] = k;
  }
  while (n_lo > 0 && n_hi > 0) {
    const int a = lo_stack[--n_lo];
    const int b = hi_stack[--n_hi];
    keep[a] = scaled[a];
    other[a] = b;
    scaled[b] = (scaled[b] + scaled[a]) - 1.0;
    if (scaled[b] < 1.0) lo_stack[n_lo++] = b; else hi_stack[n_hi++] = b;
  }
  for (int k = 0; k < NT; ++k) {
    double t = floor(keep[k] * 16777216.0 + 0.5);
    if (t > 16777215.0) t = 16777215.0;
    if (t < 0.0) t = 0.0;
    out[k] = ((uint32_t)other[k] << 24) | (uint32_t)t;
  }
}

/* One sky draw per pixel from the stream whose state is handed in (the pixel's STAGE_READ stream, between the
 * normals of two reads: wayne_amd/csrc/philox.h), advancing the state.
 * table_of[i] selects the pixel's 256-entry table, lam_level[i] is the rate the
 * table was built for, lam[i] the pixel's own rate (float32 arithmetic). */
void wayne_oracle_sky_alias_step(const float *lam, const float *lam_level, const int32_t *table_of,
                                 const uint32_t *tables, int64_t n, uint32_t *state, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    if (!(lam[i] > 0.0f)) { out[i] = 0.0; continue; }
    uint32_t *st = state + 4 * i;
    uint32_t pair[2];
    wayne_oracle_xo_next2(st, pair);      /* table word, and the first uniform of the remainder */
    const uint32_t w = pair[0];
    uint32_t wr = pair[1];
    const uint32_t col = w >> 24;
    const uint32_t entry = tables[(size_t)table_of[i] * 256 + col];
    float k = (float)(((w & 0xFFFFFFu) < (entry & 0xFFFFFFu)) ? col : (entry >> 24));
    /* the remainder in pieces of mean <= 16: Poisson variables add */
    float rest = lam[i] - lam_level[i];
    int first = 1;
    while (rest > 0.0f) {
      const float part = rest < 16.0f ? rest : 16.0f;
      rest = rest - part;
      if (!first) wr = wayne_oracle_xo_next(st);   /* further pieces: one more word each */
      first = 0;
      float u = wo_u01f(wr);
      float term = expf(-part);
      float j = 0.0f;
      for (int it = 0; it < 512 && u > term; ++it) {
        /* a term too small to move u: the uniform lies in the rounding residue of the pmf's float32 sum -- stop */
        const float left = u - term;
        if (left == u) break;
        u = left;
        j = j + 1.0f;
        term = term * (part / j);
      }
      k = k + j;
    }
    out[i] = (double)k;
  }
}

/* One Box-Muller pair per pixel from its stream's next PAIR of words (fp32, libm): the angle from the low
 * 23 bits of the first, the radius from the second. */
void wayne_oracle_normal_step(int64_t n, uint32_t *state, float *z0, float *z1) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t w[2];
    wayne_oracle_xo_next2(state + 4 * i, w);
    const float ub = wo_u01f(w[1]);
    const float R = sqrtf(-2.0f * logf(ub));
    const float ang = 6.283185307179586f * (wayne_oracle_rev12(w[0]) - 1.0f);
    z0[i] = R * cosf(ang);
    z1[i] = R * sinf(ang);
  }
}

/* Raw blocks, for uniform-integer draws (cosmic rays) and host draws. */
void wayne_oracle_philox_blocks(const uint32_t *c0, int64_t n, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t seed, uint32_t stage, uint32_t *out /* n*4 */) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t ctr[4] = {c0[i], c1, c2, c3};
    const uint32_t key[2] = {seed, stage};
    wayne_oracle_philox4x32(ctr, key, out + 4 * i);
  }
}
