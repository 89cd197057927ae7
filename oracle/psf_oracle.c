/*
 * oracle/psf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference electron thrower `PSF()`
 * (reference: wayne/pyparallel_menu.c:10-113) and of the production
 * (Philox-keyed) thrower that wayne_amd's HIP kernel implements.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path never links or calls it.
 *
 * Pinning: `wayne_oracle_psf` is checked bit-for-bit against the reference's
 * own C file compiled unmodified into oracle/_ref/libwayne_ref_psf.so
 * (oracle/Makefile), and against the golden vectors under tests/golden/
 * that were generated from that library (scripts/make_golden_psf.py).
 *
 * What the reference does (restated, not copied):
 *   A1  ssum = sum(counts)                                   (:19-34)
 *   A2  T "threads" each own electrons [t*ssum/T, (t+1)*ssum/T) (last one up
 *       to ssum), seed_t = 25234 + 17*t + test, and per electron draw
 *       theta = 2*pi*rand_r/RAND_MAX ; R = sqrt(-2 ln(rand_r/RAND_MAX));
 *       A[i] = R cos(theta) ; A[i+ssum] = R sin(theta)          (:40-64)
 *   A3  zero the frame                                        (:68-83)
 *   A4  serial scatter, bin-major: the first (int)(counts*ratio) electrons of
 *       a bin use sigma_h, the rest sigma_l; position truncates toward zero;
 *       kept iff 0 < xpos < nr and 0 < ypos < nc; pixel[ypos*nc + xpos]++
 *                                                              (:87-108)
 * The OpenMP team is emulated by looping over thread ids, so the result is
 * the one the reference produces when the runtime grants exactly `threads`
 * threads (it depends on `threads`, as the reference's does).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WO_PI 3.14159265358979323846
#define WO_RAND_MAX 2147483647

/* glibc rand_r (stdlib/rand_r.c), restated: three steps of the LCG
 * next = next*1103515245 + 12345, taking 11, 10 and 10 bits of (next>>16). */
static inline int wo_rand_r(uint32_t *state) {
  uint32_t s = *state;
  uint32_t r;
  s = s * 1103515245u + 12345u;
  r = (s >> 16) & 2047u;
  s = s * 1103515245u + 12345u;
  r = (r << 10) ^ ((s >> 16) & 1023u);
  s = s * 1103515245u + 12345u;
  r = (r << 10) ^ ((s >> 16) & 1023u);
  *state = s;
  return (int)r;
}

int wayne_oracle_rand_r(uint32_t *state) { return wo_rand_r(state); }

/* C's (int) conversion of an out-of-range / non-finite double is undefined;
 * on x86-64 (cvttsd2si) it yields INT_MIN, which the bounds test rejects.
 * Make that explicit so the oracle is well defined everywhere. */
static inline int wo_trunc_int(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
  return (int)v;
}

/* Returns 0 on success, <0 on invalid input (the reference has no checks;
 * these are the cases where it would overflow or write out of bounds). */
int wayne_oracle_psf(const int32_t *counts, int size, const double *x_pos,
                     const double *y_pos, const double *psf_ratio,
                     const double *psf_sigmal, const double *psf_sigmah,
                     int nr, int nc, int test, int threads, int32_t *out) {
  if (size < 0 || nr <= 0 || nc <= 0 || threads <= 0) return -1;
  int64_t total = 0;
  for (int k = 0; k < size; ++k) {
    if (counts[k] < 0) return -2;
    total += counts[k];
  }
  /* reference: `int ssum`, and `myid*ssum` evaluated in int (:12, :48) */
  if (total * (int64_t)threads > 2147483647LL) return -3;
  const int ssum = (int)total;

  double *A = (double *)malloc((size_t)(2 * (int64_t)ssum + 1) * sizeof(double));
  if (!A) return -4;

  for (int t = 0; t < threads; ++t) {
    int istart = t * ssum / threads;
    int iend = (t + 1) * ssum / threads;
    if (t == threads - 1) iend = ssum;
    uint32_t seed = (uint32_t)(25234 + 17 * t + test);
    for (int i = istart; i < iend; ++i) {
      double theta = 2. * WO_PI * wo_rand_r(&seed) / ((double)WO_RAND_MAX);
      double R = sqrt(-2. * log(wo_rand_r(&seed) / ((double)WO_RAND_MAX)));
      A[i] = R * cos(theta);
      A[i + ssum] = R * sin(theta);
    }
  }

  memset(out, 0, (size_t)nr * (size_t)nc * sizeof(int32_t));

  int e = 0;
  for (int b = 0; b < size; ++b) {
    const int n_wide = wo_trunc_int(counts[b] * psf_ratio[b]);
    for (int j = 0; j < counts[b]; ++j, ++e) {
      /* the first n_wide electrons of the bin take the WIDE gaussian */
      const double sig = (j < n_wide) ? psf_sigmah[b] : psf_sigmal[b];
      const int xp = wo_trunc_int(A[e] * sig + x_pos[b]);
      const int yp = wo_trunc_int(A[e + ssum] * sig + y_pos[b]);
      if (xp > 0 && xp < nr && yp > 0 && yp < nc) out[yp * nc + xp] += 1;
    }
  }
  free(A);
  return 0;
}

/* --------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123 v1.09).  The
 * reference uses numpy's MT19937 + rand_r, whose streams depend on exposure
 * order and `threads`; the MI355X path replaces them by counter-based draws
 * keyed by (seed, stage, exposure, sub-sample/read, element).  This is the
 * oracle's own statement of the generator, pinned by Random123's published
 * known-answer vectors (tests/test_philox.py).
 * ------------------------------------------------------------------------ */
void wayne_oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2],
                             uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Stage ids of the production RNG stream layout (DESIGN.md "RNG streams").
 * key = (seed, stage); counter = (element lo, element hi / draw, sub-sample or
 * read, exposure). */
enum { WO_STAGE_THROW = 2 };

/* u32 -> uniform in (0,1]: x * 2^-32 + 2^-33 as one fused multiply-add, exactly
 * as the device does (never 0; 1.0f only for x >= 0xFFFFFF80, where logf
 * gives 0 -> R = 0, a valid draw). */
static inline float wo_u01(uint32_t x) {
  return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

/* xoshiro128+ (Blackman & Vigna, 2018; public domain algorithm), the
 * oracle's own statement: 32-bit words, state seeded by one Philox block. */
typedef struct { uint32_t s[4]; } wo_xo;

static inline uint32_t wo_xo_next(wo_xo *g) {
  const uint32_t r = g->s[0] + g->s[3];
  const uint32_t t = g->s[1] << 9;
  g->s[2] ^= g->s[0];
  g->s[3] ^= g->s[1];
  g->s[1] ^= g->s[2];
  g->s[0] ^= g->s[3];
  g->s[2] ^= t;
  g->s[3] = (g->s[3] << 11) | (g->s[3] >> 21);
  return r;
}

/* Two words per state transition (device: SeededStream::next2): a = s0 + s3, the xoshiro128+ output, and
 * b = s1 + s2, the same scrambler on the other two state words, both taken BEFORE the state advances. */
static inline void wo_xo_next2(wo_xo *g, uint32_t *a, uint32_t *b) {
  *b = g->s[1] + g->s[2];
  *a = wo_xo_next(g);
}

void wayne_oracle_xo_next2(uint32_t state[4], uint32_t out[2]) {
  wo_xo g;
  memcpy(g.s, state, sizeof g.s);
  wo_xo_next2(&g, &out[0], &out[1]);
  memcpy(state, g.s, sizeof g.s);
}

/* n consecutive pairs of one stream (for the statistical tests of the pair output) */
void wayne_oracle_xo_pairs(uint32_t state[4], int64_t n, uint32_t *out /* 2n */) {
  wo_xo g;
  memcpy(g.s, state, sizeof g.s);
  for (int64_t i = 0; i < n; ++i) wo_xo_next2(&g, &out[2 * i], &out[2 * i + 1]);
  memcpy(state, g.s, sizeof g.s);
}

/* The low 23 bits of a word as a float in [1, 2): the angle of a Box-Muller draw in revolutions + 1. */
float wayne_oracle_rev12(uint32_t x) {
  const uint32_t bits = (x & 0x7fffffu) | 0x3f800000u;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint32_t wayne_oracle_xo_next(uint32_t state[4]) {
  wo_xo g;
  memcpy(g.s, state, sizeof g.s);
  uint32_t r = wo_xo_next(&g);
  memcpy(state, g.s, sizeof g.s);
  return r;
}

/*
 * Production-mode thrower for ONE sub-sample: electron e (bin-major order,
 * exactly the reference's numbering, pyparallel_menu.c:87-108) belongs to
 * block e/128; the block's stream is xoshiro128+ seeded by Philox counter
 * (e/128, 0, subsample, exposure), key (seed, STAGE_THROW); electron j = e%128
 * of the block takes the stream's j-th PAIR of words (a: angle, b: radius):
 *    u_a = low 23 bits of a / 2^23,  u_b = b 2^-32 + 2^-33,
 *    R sigma = sqrt((-2 ln2 sigma^2) log2 u_b)         (= sqrt(-2 ln u_b) sigma)
 *    x = fma(cos(2 pi u_a), R sigma, x_pos),  y = fma(sin(2 pi u_a), R sigma, y_pos)
 * in fp32 (the device uses v_sin/v_cos/v_log hardware approximations, so
 * device-vs-oracle agreement is "all but a ~1e-4 fraction of electrons land
 * in the same pixel", asserted and counted in tests/test_psf_gpu.py).
 * x_pos, y_pos above are the FRACTION of the bin's pixel: the fp64 position is split
 * once per bin into floor(pos) and (float)(pos - floor(pos)), the float32 offset is
 * added to the fraction and the electron's pixel is floor(pos) + floor(that sum)
 * (wayne_amd/csrc/common.h, bin_local; split_oracle.c, so_bin_local) -- floor where
 * the reference truncates (:91-92), which keeps the same electrons under the strict
 * bounds 0 < pos < n (:93).  Same sigma split and bounds rule as A4.
 */
static int wo_cell(int o, float v) {
  if (!(v > -2147483904.0f)) return INT32_MIN;
  if (!(v < 2147483648.0f)) return INT32_MAX;
  const int64_t c = (int64_t)o + (int64_t)floorf(v);
  return c < INT32_MIN ? INT32_MIN : (c > INT32_MAX ? INT32_MAX : (int)c);
}

int wayne_oracle_psf_philox(const int32_t *counts, int size,
                            const double *x_pos, const double *y_pos,
                            const double *psf_ratio, const float *psf_sigmal,
                            const float *psf_sigmah, int nr, int nc,
                            uint32_t seed, uint32_t exposure,
                            uint32_t subsample, int32_t *out) {
  if (size < 0 || nr <= 0 || nc <= 0) return -1;
  memset(out, 0, (size_t)nr * (size_t)nc * sizeof(int32_t));
  const uint32_t key[2] = {seed, WO_STAGE_THROW};
  uint64_t e = 0;
  wo_xo g = {{0, 0, 0, 0}};
  for (int b = 0; b < size; ++b) {
    if (counts[b] < 0) return -2;
    /* sigma split in fp64, as the reference (:89) */
    const int n_wide = wo_trunc_int(counts[b] * psf_ratio[b]);
    const int sane = fabs(x_pos[b]) < 1e6 && fabs(y_pos[b]) < 1e6;
    const double flx = sane ? floor(x_pos[b]) : 0.0, fly = sane ? floor(y_pos[b]) : 0.0;
    const float fx = (float)(x_pos[b] - flx), fy = (float)(y_pos[b] - fly);
    const int ox = (int)flx, oy = (int)fly;
    for (int j = 0; j < counts[b]; ++j, ++e) {
      if ((e & 127u) == 0) {
        /* new block of 128 electrons: fresh stream */
        const uint32_t ctr[4] = {(uint32_t)(e >> 7), 0u, subsample, exposure};
        wayne_oracle_philox4x32(ctr, key, g.s);
      }
      uint32_t ra, rb;
      wo_xo_next2(&g, &ra, &rb);
      const float ang = 6.283185307179586f * (wayne_oracle_rev12(ra) - 1.0f);
      const float sig = (j < n_wide) ? psf_sigmah[b] : psf_sigmal[b];
      const float c = (-1.3862943611198906f * sig) * sig;
      if (!(c > -3e38f)) continue;          /* a sigma that is not finite: none of its electrons is kept (:91-93) */
      const float Rs = sqrtf(c * log2f(wo_u01(rb)));
      const int xp = wo_cell(ox, fmaf(cosf(ang), Rs, fx));
      const int yp = wo_cell(oy, fmaf(sinf(ang), Rs, fy));
      if (xp > 0 && xp < nr && yp > 0 && yp < nc) out[yp * nc + xp] += 1;
    }
  }
  return 0;
}
