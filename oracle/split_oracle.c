/*
 * oracle/split_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU statement of the production thrower's default mode (WAYNE_RNG_SPLIT,
 * include/wayne_hip.h): what the reference does electron by electron
 * (pyparallel_menu.c:87-108) is, for the narrow gaussian of a well-populated
 * bin, drawn as ONE multinomial over the pixels around the bin.
 *
 * Throwing n electrons independently at pixels with probabilities p_ij gives
 * multinomial(n; p_ij) pixel counts; x and y of the reference's electron are
 * independent normals (:57-61, :99-107), so p_ij = P_i Q_j with P, Q the
 * masses of N(pos, sigma_l^2) on [i, i+1) -- the reference's (int) truncation
 * is floor() wherever it keeps the electron (:91-93).  The multinomial is
 * sampled as a chain of conditional binomials: columns centre-out, then the
 * rows of each non-empty column centre-out, over a +-6 pixel window
 * (>= 6.5 sigma_l; the < 1e-10 of mass outside the window is not thrown).
 *
 * Parity status: the reference has no such mode -- its result for the same
 * inputs is ONE sample of the same distribution -- so this file is "parity
 * unpinned" against the reference and pinned instead by
 *   (a) distribution tests against scipy.stats.binom (tests/test_samplers.py),
 *   (b) agreement with the device on the same counters (tests/test_split_gpu.py),
 *   (c) the moments of the per-electron thrower, which IS pinned bit-for-bit.
 * Binomial(n, p): inversion by sequential search below n*min(p,1-p) = 10
 * (Kachitvichyanukul & Schmeiser 1988, BINV), Hoermann's BTRS transformed
 * rejection above (J. Stat. Comput. Simul. 46, 1993).  fp32 arithmetic, libm,
 * no contraction (Makefile: -ffp-contract=off), like the device's
 * "exact sampler" policy.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

void wayne_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t wayne_oracle_xo_next(uint32_t state[4]);
void wayne_oracle_xo_next2(uint32_t state[4], uint32_t out[2]);
float wayne_oracle_rev12(uint32_t x);

enum { SO_STAGE_THROW = 2, SO_STAGE_NARROW = 9, SO_STAGE_LANE = 10, SO_WINDOW = 6, SO_CELLS = 2 * SO_WINDOW + 1 };

static float so_u01(uint32_t x) { return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f); }

/* ln k! minus its Stirling approximation sqrt(2 pi) (k+1)^(k+1/2) e^-(k+1) */
static float so_fc(float k) {
  static const float small[10] = {
      0.0810614667953272f,  0.0413406959554092f,  0.0276779256849983f, 0.02079067210376509f,
      0.0166446911898211f,  0.0138761288230707f,  0.0118967099458917f, 0.0104112652619720f,
      0.00925546218271273f, 0.00833056343336287f};
  if (k < 10.0f) return small[(int)k];
  const float k1 = k + 1.0f, k1s = k1 * k1;
  return ((float)(1.0 / 12) - ((float)(1.0 / 360) - (float)(1.0 / 1260) / k1s) / k1s) / k1;
}

static float so_binomial(float n, float p, uint32_t state[4]);

/* Diagnostic hook (scripts/diagnose_split_flip.py): while a trace buffer is set, every binomial call is recorded
 * as (n, p, result), and call number `perturb_at` has its p multiplied by (1 + perturb_rel) -- to find the draw of
 * a chain whose outcome hangs on the last bits of its probability. */
typedef struct { float n, p, x; } so_call;
static so_call *so_trace_buf = 0;
static int so_trace_cap = 0, so_trace_len = 0, so_perturb_at = -1;
static float so_perturb_rel = 0.0f;
void wayne_oracle_binomial_trace(float *buf3, int cap, int perturb_at, float perturb_rel) {
  so_trace_buf = (so_call *)buf3; so_trace_cap = cap; so_trace_len = 0;
  so_perturb_at = perturb_at; so_perturb_rel = perturb_rel;
}
int wayne_oracle_binomial_trace_len(void) { return so_trace_len; }

float wayne_oracle_binomial_f(float n, float p, uint32_t state[4]) {
  if (!so_trace_buf) return so_binomial(n, p, state);
  const int i = so_trace_len++;
  if (i == so_perturb_at) p = p * (1.0f + so_perturb_rel);
  const float x = so_binomial(n, p, state);
  if (i < so_trace_cap) { so_trace_buf[i].n = n; so_trace_buf[i].p = p; so_trace_buf[i].x = x; }
  return x;
}

static float so_binomial(float n, float p, uint32_t state[4]) {
  if (!(n > 0.0f) || !(p > 0.0f)) return 0.0f;
  if (p >= 1.0f) return n;
  const int mirrored = p > 0.5f;
  if (mirrored) p = 1.0f - p;
  const float q = 1.0f - p;
  float x = 0.0f;
  if (n * p < 10.0f) {
    /* BINV: f(0) = q^n, f(x) = f(x-1) * ((n+1) s / x - s), s = p/q */
    const float s = p / q;
    const float a = (n + 1.0f) * s;
    float f = expf(n * log1pf(-p));
    float u = so_u01(wayne_oracle_xo_next(state));
    /* 64 steps: unreachable for n p < 10 unless the uniform lies in the rounding residue of the pmf's float sum
     * (~6e-8 of the draws) -- then the mean is returned */
    for (int it = 0; it < 64 && u > f; ++it) {
      u = u - f;
      x = x + 1.0f;
      f = f * (a / x - s);
      if (x >= n) { x = n; break; }
    }
    if (x >= 64.0f && x < n) x = floorf(n * p + 0.5f);
  } else {
    /* BTRS */
    const float spq = sqrtf(n * p * q);
    const float b = 1.15f + 2.53f * spq;
    const float a = -0.0873f + 0.0248f * b + 0.01f * p;
    const float c = n * p + 0.5f;
    const float vr = 0.92f - 4.2f / b;
    const float alpha = (2.83f + 5.1f / b) * spq;
    const float m = floorf((n + 1.0f) * p);
    const float r = p / q;
    const float nm = n - m + 1.0f;
    /* the part of the acceptance bound that the trial does not change */
    const float fixed = (m + 0.5f) * logf((m + 1.0f) / (r * nm)) + so_fc(m) + so_fc(n - m);
    x = floorf(n * p + 0.5f);
    for (int it = 0; it < 256; ++it) {
      uint32_t w[2];
      wayne_oracle_xo_next2(state, w);        /* one pair of words per trial */
      const float U = so_u01(w[0]) - 0.5f;
      const float V = so_u01(w[1]);
      const float us = 0.5f - fabsf(U);
      const float k = floorf((2.0f * a / us + b) * U + c);
      if (us >= 0.07f && V <= vr) { x = k; break; }
      if (k < 0.0f || k > n) continue;
      const float v = logf(V * alpha / (a / (us * us) + b));
      const float nk = n - k + 1.0f;
      const float bound = fixed + (n + 1.0f) * logf(nm / nk) + (k + 0.5f) * logf(r * nk / (k + 1.0f)) - so_fc(k) -
                          so_fc(n - k);
      if (v <= bound) { x = k; break; }
    }
  }
  return mirrored ? n - x : x;
}

/* n draws of Binomial(n[i], p[i]), element i from the STAGE_NARROW stream of "bin" i. */
void wayne_oracle_binomial_vec(const float *n, const float *p, int64_t count, uint32_t seed,
                               uint32_t subsample, uint32_t exposure, float *out) {
  const uint32_t key[2] = {seed, SO_STAGE_NARROW};
  for (int64_t i = 0; i < count; ++i) {
    const uint32_t ctr[4] = {(uint32_t)i, 0u, subsample, exposure};
    uint32_t st[4];
    wayne_oracle_philox4x32(ctr, key, st);
    out[i] = wayne_oracle_binomial_f(n[i], p[i], st);
  }
}

/* P(Z > t), t >= 0, as k_narrow.h upper_tail: the Chebyshev fit of erfc (Numerical Recipes erfcc, fractional
 * error < 1.2e-7 before rounding) in float, nothing beyond 6.5 sigma.  tests/test_samplers.py checks it against
 * scipy's erfc. */
float wayne_oracle_upper_tail(float t) {
  if (t > 6.5f) return 0.0f;
  const float z = t * 0.70710678118654752f;
  const float u = 1.0f / (1.0f + 0.5f * z);
  float p = 0.17087277f;
  p = fmaf(p, u, -0.82215223f);
  p = fmaf(p, u, 1.48851587f);
  p = fmaf(p, u, -1.13520398f);
  p = fmaf(p, u, 0.27886807f);
  p = fmaf(p, u, -0.18628806f);
  p = fmaf(p, u, 0.09678418f);
  p = fmaf(p, u, 0.37409196f);
  p = fmaf(p, u, 1.00002368f);
  p = fmaf(p, u, -1.26551223f);
  p = fmaf(-z, z, p);
  return (0.5f * u) * expf(p);
}
static float so_tail(float t) { return wayne_oracle_upper_tail(t); }

/* Masses of N(frac, sigma^2) on the 13 unit cells of the window, in visiting
 * order: centre, +1, -1, +2, -2, ...  `before[c]` = mass not yet visited when
 * cell c is reached (1 for the centre, the two remaining tails afterwards). */
static void so_cell_masses(float frac, float inv_sigma, float mass[SO_CELLS], float before[SO_CELLS]) {
  float above = so_tail((1.0f - frac) * inv_sigma);
  float below = so_tail(frac * inv_sigma);
  mass[0] = 1.0f - above - below;
  before[0] = 1.0f;
  for (int c = 1; c < SO_CELLS; ++c) {
    const int d = (c + 1) / 2;
    before[c] = above + below;
    if (c % 2) {
      const float next = so_tail(((float)(d + 1) - frac) * inv_sigma);
      mass[c] = above - next;
      above = next;
    } else {
      const float next = so_tail(((float)d + frac) * inv_sigma);
      mass[c] = below - next;
      below = next;
    }
  }
}

static int so_cell_offset(int c) { return c == 0 ? 0 : (c % 2 ? (c + 1) / 2 : -(c / 2)); }

static float so_clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

/* Bin-local coordinates (wayne_amd/csrc/common.h, bin_local): a bin's position is split in fp64 into its pixel o and
 * the fraction f of it, pos = o + f, f = (float)(pos - o); an electron's float32 offset is added to f and its pixel is
 * o + floor(f + offset).  floor, where the reference truncates toward zero (pyparallel_menu.c:91-92): the two differ
 * only for frame coordinates in (-1, 0), which both leave outside 0 < pos < n (:93).  A position that is not finite or
 * beyond +-1e6 is not split: o = 0, f = (float)pos. */
typedef struct { float fx, fy; int ox, oy; int sane; } so_local;

static so_local so_bin_local(double xd, double yd) {
  so_local b;
  b.sane = fabs(xd) < 1e6 && fabs(yd) < 1e6;
  const double flx = b.sane ? floor(xd) : 0.0, fly = b.sane ? floor(yd) : 0.0;
  b.fx = (float)(xd - flx); b.fy = (float)(yd - fly);
  b.ox = (int)flx; b.oy = (int)fly;
  return b;
}

/* o + floor(v); a NaN or a value beyond int32 lands off every frame (the device saturates the conversion) */
static int so_cell(int o, float v) {
  if (!(v > -2147483904.0f)) return INT32_MIN;
  if (!(v < 2147483648.0f)) return INT32_MAX;
  const int64_t c = (int64_t)o + (int64_t)floorf(v);
  return c < INT32_MIN ? INT32_MIN : (c > INT32_MAX ? INT32_MAX : (int)c);
}

/*
 * The whole thrower in split mode for one sub-sample.  Bins with at least
 * `split_min` narrow electrons (and 0.05 < sigma_l <= 6/6.5) hand them to the
 * multinomial.  What is left to throw one by one -- the wide electrons of such a
 * bin, or the whole of a bin that does not qualify -- is thrown from the bin's
 * OWN stream when it is at most `lane_max` electrons (the device gives each bin a
 * lane, k_lane): electron j takes WORD j of the xoshiro128+ stream seeded by
 * Philox block (bin, 0, sub-sample, exposure), stage LANE.  Beyond `lane_max`
 * the electrons are numbered bin-major over the bins thrown that way and drawn
 * from the STAGE_THROW block streams exactly as wayne_oracle_psf_philox does.
 * Either way: the first n_wide electrons of a bin take sigma_h
 * (pyparallel_menu.c:89-107), same arithmetic as wayne_oracle_psf_philox.
 */
static void so_throw_one(uint32_t g[4], const so_local *L, float sig, int n, int32_t *out) {
  uint32_t w[2];
  wayne_oracle_xo_next2(g, w);            /* one pair per electron: angle, radius */
  const float ang = 6.283185307179586f * (wayne_oracle_rev12(w[0]) - 1.0f);
  const float c = (-1.3862943611198906f * sig) * sig;
  if (!(c > -3e38f)) return;                /* a sigma that is not finite: the reference keeps none of its electrons (:91-93) */
  const float Rs = sqrtf(c * log2f(so_u01(w[1])));
  const int xp = so_cell(L->ox, fmaf(cosf(ang), Rs, L->fx));
  const int yp = so_cell(L->oy, fmaf(sinf(ang), Rs, L->fy));
  if (xp > 0 && xp < n && yp > 0 && yp < n) out[(size_t)yp * n + xp] += 1;
}

/* An electron of a bin's own lane (stage LANE): ONE word of the bin's stream -- its high 23 bits the angle (as the
 * mantissa of a float in [1, 2) revolutions), its low half h the radius, u = (h + 1/2) / 2^16; h = 0 (the far tail,
 * where the cell is not small) is subdivided by 17 bits of the bin's side LCG: u = (h' + 1/2) / 2^33.  Same arithmetic
 * as the device's k_lane (k_narrow.h), libm for the hardware's log2 / sqrt / sin / cos. */
static void so_throw_word(uint32_t wd, uint32_t *refine, const so_local *L, float sig, int n, int32_t *out) {
  const uint32_t bits = 0x3f800000u | (wd >> 9);
  float rev;
  memcpy(&rev, &bits, 4);
  const float ang = 6.283185307179586f * (rev - 1.0f);
  const float c = (-1.3862943611198906f * sig) * sig;
  const uint32_t h = wd & 0xFFFFu;
  if (h == 0u) *refine = *refine * 1664525u + 1013904223u;    /* (the side stream advances whether or not the electron is kept) */
  if (!(c > -3e38f)) return;                /* a sigma that is not finite: the reference keeps none of its electrons (:91-93) */
  float r2 = fmaf(c, log2f((float)h + 0.5f), -16.0f * c);
  if (h == 0u) r2 = fmaf(c, log2f((float)(*refine >> 15) + 0.5f), -33.0f * c);
  const float Rs = sqrtf(r2);
  const int xp = so_cell(L->ox, fmaf(cosf(ang), Rs, L->fx));
  const int yp = so_cell(L->oy, fmaf(sinf(ang), Rs, L->fy));
  if (xp > 0 && xp < n && yp > 0 && yp < n) out[(size_t)yp * n + xp] += 1;
}

/*
 * Pooled rows (k_narrow.h, "row pooling").  16 consecutive bins of a sub-sample sit at nearly the same height
 * y and have nearly the same sigma_l, so their row distributions q_b(t) over a common window of SO_PROWS absolute
 * rows are nearly equal.  Any electron's row can be drawn from the mixture
 *     q_b = Z * (qbar / Z) + (1 - Z) * r_b,     qbar(t) = min_b q_b(t),  Z = sum_t qbar(t),
 *     r_b = (q_b - qbar) / (1 - Z)
 * -- with probability Z from the distribution the whole group shares, otherwise from the bin's own residual --
 * which is exact for every Z in (0, 1].  Columns and rows are independent, so: each bin splits its electrons into
 * "common" (Binomial(n, Z)) and "residual" ones; the common ones go through the bin's column chain and are POOLED
 * per absolute column over the group; each column then needs ONE row chain for the whole group (its own stream,
 * stage POOL); the residual electrons (a fraction of a per cent) are thrown one by one, column from the gaussian
 * itself and row by inverse CDF on r_b.  A group is pooled when it has at least two split bins that are close
 * enough (so_group_pools); otherwise every bin runs its own column and row chains as before.
 */
enum { SO_GROUP = 16, SO_PROWS = 2 * SO_WINDOW + 2, SO_STAGE_POOL = 11 };

/* masses of N(y, sigma^2) on the rows [J0 + t, J0 + t + 1), t = 0..SO_PROWS-1, normalised to sum 1, from
 * yl = y - J0 (the bin's height above the bottom row of the window); every mass is a difference of two tails on the
 * same side of y (no cancellation), the row that holds y is 1 - both tails */
static void so_rows_abs(float yl, float inv_s, float q[SO_PROWS]) {
  float A[SO_PROWS + 1], E[SO_PROWS + 1];
  for (int t = 0; t <= SO_PROWS; ++t) {
    E[t] = (float)t - yl;
    A[t] = so_tail(fabsf(E[t]) * inv_s);
  }
  float S = 0.0f;
  for (int t = 0; t < SO_PROWS; ++t) {
    float m;
    if (E[t] >= 0.0f) m = A[t] - A[t + 1];
    else if (E[t + 1] <= 0.0f) m = A[t + 1] - A[t];
    else m = 1.0f - A[t] - A[t + 1];
    q[t] = m > 0.0f ? m : 0.0f;
    S += q[t];
  }
  for (int t = 0; t < SO_PROWS; ++t) q[t] = q[t] / S;
}

/* a bin of a group: pixel (ic, jc) + fraction (fx, fy) of its position (so_bin_local); yd: the position's y in fp64 */
typedef struct { int split, sane; float n, fx, fy, sl; double yd; int ic, jc; } so_bin;

/* the bin's height above row J0, rounded once from fp64 */
static float so_yl(const so_bin *b, int J0) { return (float)(b->yd - (double)J0); }

static int so_group_pools(const so_bin *B, int nb) {
  int act = 0, ic0 = INT32_MAX, ic1 = INT32_MIN, jc0 = INT32_MAX;
  float s0 = 3e38f, s1 = 0.0f;
  int64_t tot = 0;
  for (int i = 0; i < nb; ++i) {
    if (!B[i].split) continue;
    if (!B[i].sane) return 0;
    ++act;
    if (B[i].ic < ic0) ic0 = B[i].ic;
    if (B[i].ic > ic1) ic1 = B[i].ic;
    if (B[i].jc < jc0) jc0 = B[i].jc;
    s0 = fminf(s0, B[i].sl); s1 = fmaxf(s1, B[i].sl);
    tot += (int64_t)B[i].n;
  }
  if (act < 2) return 0;
  float y0 = 3e38f, y1 = -3e38f;
  for (int i = 0; i < nb; ++i) {
    if (!B[i].split) continue;
    const float yl = so_yl(&B[i], jc0 - SO_WINDOW);
    y0 = fminf(y0, yl); y1 = fmaxf(y1, yl);
  }
  return ic1 - ic0 <= 2 && (y1 - y0) <= 0.25f * s0 && s1 <= 1.1f * s0 && tot <= 16777216;
}

static void so_put(int32_t *out, int n, int col, int row, float m) {
  if (m > 0.0f && col > 0 && col < n && row > 0 && row < n) out[(size_t)row * n + col] += (int32_t)m;
}

static void so_narrow_own(const so_bin *b, int bin, const uint32_t key_n[2], uint32_t subsample, uint32_t exposure,
                          int n, int32_t *out) {
  if (!b->sane) return;          /* (none of its electrons can reach the frame: k_narrow.h) */
  const float inv_s = 1.0f / b->sl;
  float P[SO_CELLS], Pb[SO_CELLS], Q[SO_CELLS], Qb[SO_CELLS];
  so_cell_masses(b->fx, inv_s, P, Pb);
  so_cell_masses(b->fy, inv_s, Q, Qb);
  const uint32_t ctr[4] = {(uint32_t)bin, 0u, subsample, exposure};
  uint32_t st[4];
  wayne_oracle_philox4x32(ctr, key_n, st);
  float left = b->n;
  for (int c = 0; c < SO_CELLS && left > 0.0f; ++c) {
    const float in_col = wayne_oracle_binomial_f(left, so_clamp01(P[c] / Pb[c]), st);
    left -= in_col;
    const int col = b->ic + so_cell_offset(c);
    float col_left = in_col;
    for (int r = 0; r < SO_CELLS && col_left > 0.0f; ++r) {
      /* row r given that none of the rows before it was hit: its mass over the two tails still unvisited */
      const float m = wayne_oracle_binomial_f(col_left, so_clamp01(Q[r] / Qb[r]), st);
      col_left -= m;
      so_put(out, n, col, b->jc + so_cell_offset(r), m);
    }
  }
}

static void so_narrow_pooled(const so_bin *B, int nb, int bin0, uint32_t seed, uint32_t subsample,
                             uint32_t exposure, int n, int32_t *out) {
  const uint32_t key_n[2] = {seed, SO_STAGE_NARROW};
  const uint32_t key_p[2] = {seed, SO_STAGE_POOL};
  int ic0 = INT32_MAX, jc0 = INT32_MAX;
  for (int i = 0; i < nb; ++i)
    if (B[i].split) { if (B[i].ic < ic0) ic0 = B[i].ic; if (B[i].jc < jc0) jc0 = B[i].jc; }
  const int X0 = ic0 - SO_WINDOW, J0 = jc0 - SO_WINDOW;
  float q[SO_GROUP][SO_PROWS], qbar[SO_PROWS];
  for (int t = 0; t < SO_PROWS; ++t) qbar[t] = 3e38f;
  for (int i = 0; i < nb; ++i) {
    if (!B[i].split) continue;
    so_rows_abs(so_yl(&B[i], J0), 1.0f / B[i].sl, q[i]);
    for (int t = 0; t < SO_PROWS; ++t) qbar[t] = fminf(qbar[t], q[i][t]);
  }
  /* tails of qbar about the centre row SO_WINDOW, summed from the far ends inwards */
  float PL[SO_PROWS], SU[SO_PROWS + 1];
  PL[0] = qbar[0];
  for (int t = 1; t <= SO_WINDOW; ++t) PL[t] = PL[t - 1] + qbar[t];
  SU[SO_PROWS] = 0.0f;
  for (int t = SO_PROWS - 1; t > SO_WINDOW; --t) SU[t] = SU[t + 1] + qbar[t];
  const float Z = fminf(PL[SO_WINDOW] + SU[SO_WINDOW + 1], 1.0f);

  float pooled[SO_GROUP];
  for (int j = 0; j < SO_GROUP; ++j) pooled[j] = 0.0f;
  for (int i = 0; i < nb; ++i) {
    if (!B[i].split) continue;
    const so_bin *b = &B[i];
    const uint32_t ctr[4] = {(uint32_t)(bin0 + i), 0u, subsample, exposure};
    uint32_t st[4];
    wayne_oracle_philox4x32(ctr, key_n, st);
    const float common = wayne_oracle_binomial_f(b->n, Z, st);
    const int residual = (int)(b->n - common);
    /* the common electrons' columns */
    const float inv_s = 1.0f / b->sl;
    float P[SO_CELLS], Pb[SO_CELLS];
    so_cell_masses(b->fx, inv_s, P, Pb);
    float left = common;
    for (int c = 0; c < SO_CELLS && left > 0.0f; ++c) {
      const float in_col = wayne_oracle_binomial_f(left, so_clamp01(P[c] / Pb[c]), st);
      left -= in_col;
      pooled[b->ic + so_cell_offset(c) - X0] += in_col;
    }
    /* the residual electrons, one by one: two pairs each (column from the gaussian, row by inverse CDF on q - qbar) */
    float D[SO_PROWS], R = 0.0f;                /* running sums of the residual row masses */
    for (int t = 0; t < SO_PROWS; ++t) { R += q[i][t] - qbar[t]; D[t] = R; }
    const float cs = (-1.3862943611198906f * b->sl) * b->sl;
    for (int e = 0; e < residual; ++e) {
      uint32_t w[2], v[2];
      wayne_oracle_xo_next2(st, w);
      wayne_oracle_xo_next2(st, v);
      const float ang = 6.283185307179586f * (wayne_oracle_rev12(w[0]) - 1.0f);
      const float Rs = sqrtf(cs * log2f(so_u01(w[1])));
      int col = so_cell(0, fmaf(cosf(ang), Rs, b->fx));
      if (col < -SO_WINDOW) col = -SO_WINDOW;
      if (col > SO_WINDOW) col = SO_WINDOW;
      col += b->ic;
      const float u = so_u01(v[0]) * R;
      int row = 0;
      for (int t = 0; t < SO_PROWS - 1; ++t) row += (u >= D[t]);
      so_put(out, n, col, J0 + row, 1.0f);
    }
  }
  /* one row chain per pooled column: centre row first, then alternately above and below */
  for (int j = 0; j < SO_GROUP; ++j) {
    if (!(pooled[j] > 0.0f)) continue;
    const uint32_t ctr[4] = {(uint32_t)(bin0 / SO_GROUP), (uint32_t)j, subsample, exposure};
    uint32_t st[4];
    wayne_oracle_philox4x32(ctr, key_p, st);
    float left = pooled[j];
    int up = SO_WINDOW + 1, lo = SO_WINDOW - 1;
    for (int i = 0; i < SO_PROWS && left > 0.0f; ++i) {
      int t;
      float rem;
      if (i == 0) { t = SO_WINDOW; rem = PL[SO_WINDOW] + SU[SO_WINDOW + 1]; }
      else if (i & 1) { t = up; rem = SU[up] + (lo >= 0 ? PL[lo] : 0.0f); ++up; }
      else { t = lo; rem = SU[up] + PL[lo]; --lo; }
      const float m = wayne_oracle_binomial_f(left, so_clamp01(qbar[t] / rem), st);
      left -= m;
      so_put(out, n, X0 + j, J0 + t, m);
    }
  }
}

int wayne_oracle_psf_split(const int32_t *counts, int size, const double *x_pos, const double *y_pos,
                           const double *psf_ratio, const double *psf_sigmal, const double *psf_sigmah,
                           int n, int split_min, int lane_max, uint32_t seed, uint32_t exposure,
                           uint32_t subsample, int32_t *out) {
  if (size < 0 || n <= 0) return -1;
  memset(out, 0, (size_t)n * (size_t)n * sizeof(int32_t));
  const uint32_t key_t[2] = {seed, SO_STAGE_THROW};
  const uint32_t key_n[2] = {seed, SO_STAGE_NARROW};
  const uint32_t key_l[2] = {seed, SO_STAGE_LANE};
  uint64_t e = 0;
  uint32_t g[4] = {0, 0, 0, 0};
  for (int b0 = 0; b0 < size; b0 += SO_GROUP) {
    const int nb = size - b0 < SO_GROUP ? size - b0 : SO_GROUP;
    so_bin B[SO_GROUP];
    for (int i = 0; i < nb; ++i) {
      const int b = b0 + i;
      if (counts[b] < 0) return -2;
      const double nwd = (double)counts[b] * psf_ratio[b];
      int64_t n_wide = (nwd >= 2147483647.0) ? 2147483647 : (nwd > -2147483648.0 ? (int64_t)(int32_t)nwd : -2147483648LL);
      if (n_wide < 0) n_wide = 0;
      if (n_wide > counts[b]) n_wide = counts[b];
      const int64_t n_narrow = counts[b] - n_wide;
      /* (the chain counts in float32: bins beyond 2^24 narrow electrons stay with the per-electron thrower) */
      const int split = split_min > 0 && n_narrow >= split_min && n_narrow <= 16777216 && psf_sigmal[b] > 0.05 &&
                        psf_sigmal[b] * 6.5 <= (double)SO_WINDOW;
      const so_local L = so_bin_local(x_pos[b], y_pos[b]);
      const float sl = (float)psf_sigmal[b], sh = (float)psf_sigmah[b];
      B[i].split = split; B[i].sane = L.sane; B[i].n = (float)n_narrow; B[i].fx = L.fx; B[i].fy = L.fy; B[i].sl = sl;
      B[i].yd = y_pos[b]; B[i].ic = L.ox; B[i].jc = L.oy;

      /* one by one */
      const int64_t thrown = split ? n_wide : counts[b];
      if (split_min > 0 && thrown <= lane_max) {
        const uint32_t ctr[4] = {(uint32_t)b, 0u, subsample, exposure};
        uint32_t gl[4];
        wayne_oracle_philox4x32(ctr, key_l, gl);
        uint32_t refine = (gl[1] * 0x9E3779B9u) ^ gl[3], w2[2] = {0u, 0u};
        for (int64_t j = 0; j < thrown; ++j) {
          if ((j & 1) == 0) wayne_oracle_xo_next2(gl, w2);          /* a pair of the stream serves two electrons */
          so_throw_word(w2[j & 1], &refine, &L, (j < n_wide) ? sh : sl, n, out);
        }
      } else {
        for (int64_t j = 0; j < thrown; ++j, ++e) {
          if ((e & 127u) == 0) {
            const uint32_t ctr[4] = {(uint32_t)(e >> 7), 0u, subsample, exposure};
            wayne_oracle_philox4x32(ctr, key_t, g);
          }
          so_throw_one(g, &L, (j < n_wide) ? sh : sl, n, out);
        }
      }
    }
    /* the narrow component of the group's split bins as multinomials */
    if (so_group_pools(B, nb)) {
      so_narrow_pooled(B, nb, b0, seed, subsample, exposure, n, out);
    } else {
      for (int i = 0; i < nb; ++i)
        if (B[i].split) so_narrow_own(&B[i], b0 + i, key_n, subsample, exposure, n, out);
    }
  }
  return 0;
}
