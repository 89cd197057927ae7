// Philox4x32-10 counter-based RNG and the uniform / normal transforms built
// on it.  Replaces the reference's order-dependent numpy MT19937 stream
// (reference: wayne/run_visit.py:68-77, exposure_generator.py:327-329,495,626,
// detector.py:191,198, cosmic_rays.py:80-81,127,134) and its per-thread
// rand_r streams (pyparallel_menu.c:52-58) with draws that are a pure function
// of (seed, stage, exposure, sub-sample | read, element), so results do not
// depend on how exposures are sharded over GPUs or on launch geometry.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define WAYNE_HD __host__ __device__ __forceinline__
#else
#define WAYNE_HD inline
#endif

namespace wayne {

// RNG stages: key = (seed, stage); counter = (c0, c1, c2, c3) as commented.
// "Philox stream" stages read consecutive Philox blocks (c1 = block index).
// "seeded stream" stages use ONE Philox block as the 128-bit state of a short
// xoshiro128+ sequence (Blackman & Vigna 2018): the bulk draws -- one stream
// per 128 thrown electrons, one per pixel -- cost ~8 full-rate VALU ops per
// word instead of Philox's five quarter-rate 32x32->64 multiplies, and are
// still a pure function of (seed, stage, exposure, element): independent of
// sharding and launch geometry.
enum Stage : uint32_t {
  STAGE_COUNTS = 1,   // Philox stream (bin w, block, sub-sample k, exposure)     stellar Poisson
  STAGE_THROW = 2,    // seeded stream (electron block e>>7, 0, sub-sample k, exposure): pair j (next2) -> electron j of the block
  STAGE_SKY = 3,      // seeded stream (pixel, 0, 0, exposure): sky Poisson draws of reads 0..R-1 in order (a pair per trial) -- the
                      //   DIRECT sampler only (k_ramp SKY = 0: rates that fit no alias table); the table-driven draw reads STAGE_READ
  STAGE_CR_COUNT = 4, // Philox stream (0, block, read r, exposure)               number of cosmic hits
  STAGE_CR_HIT = 5,   // Philox block  (hit i, 0, read r, exposure)               energy, y, x of hit i
  STAGE_READ = 6,     // seeded stream (pixel, 0, 0, exposure): the pair of (dark, read-noise) normals of read 0 (the zero read), then
                      //   for r = 0..R-1: the words of read interval r's table-driven sky draw (a pair: table word + remainder
                      //   uniform, further pieces one word each; none where the pixel's sky rate is 0) and the pair of read r+1's normals
  STAGE_NOISE = 7,    // seeded stream (pixel, 0, 0, exposure): pair r -> optional gaussian noise of read interval r
  STAGE_HOST = 8,     // Philox block  (sub-sample k, 0, 0, exposure)             jitter x/y, replay seed
  STAGE_NARROW = 9,   // seeded stream (bin w, 0, sub-sample k, exposure): the binomial chain that splits a bin's
                      //   narrow-PSF electrons over pixels (k_narrow, rng_mode WAYNE_RNG_SPLIT)
  STAGE_LANE = 10,    // seeded stream (bin w, 0, sub-sample k, exposure): WORD j (pair j / 2) -> electron j of the electrons a
                      //   bin's own lane throws one by one (k_lane, rng_mode WAYNE_RNG_SPLIT): the wide-PSF electrons of a bin
                      //   whose narrow ones went to the multinomial, or every electron of a thinly populated bin (wide ones
                      //   first); an electron whose radius half-word is 0 refines it from a side LCG seeded by the same block (k_lane)
  STAGE_POOL = 11,    // seeded stream (bin group w >> 4, column j of the group's window, sub-sample k, exposure): the row chain
                      //   of the electrons the 16 bins of a group put into that column (k_narrow, pooled rows)
};
constexpr uint32_t kThrowBlock = 128;   // electrons per STAGE_THROW stream

struct u32x4 {
  uint32_t v[4];
};

WAYNE_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                             uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  u32x4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// uint32 -> uniform in (0, 1]: x * 2^-32 + 2^-33 as ONE fused multiply-add
// (cvt + fma on the device), fp32 / fp64.
WAYNE_HD float u01f(uint32_t x) { return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f); }
WAYNE_HD double u01d(uint32_t x) { return fma((double)x, 2.3283064365386963e-10, 1.1641532182693481e-10); }

// uniform integer in [0, n) by multiply-shift (bias <= n / 2^32).
WAYNE_HD uint32_t uint_below(uint32_t x, uint32_t n) {
  return (uint32_t)(((uint64_t)x * (uint64_t)n) >> 32);
}

// A sequential reader of one Philox stream: counter (c0, block, c2, c3) with
// the block index advancing every four words.
struct PhiloxStream {
  uint32_t c0, c2, c3, k0, k1, block, have;
  u32x4 buf;
#ifdef WAYNE_NEGCTL_ADDITIVE_KEY
  // NEGATIVE-CONTROL builds of tests/test_independence_gpu.py (never the shipped library): element, sub-sample / read and
  // exposure index ADDED into one counter word -- (element e, exposure i + 1) then shares its stream with (e + 1, i), and
  // (sub-sample 1, exposure 0) with (0, 1): marginal laws untouched, independence gone
  WAYNE_HD PhiloxStream(uint32_t seed, uint32_t stage, uint32_t c0_, uint32_t c2_,
                        uint32_t c3_)
      : c0(c0_ + c2_ + c3_), c2(0u), c3(0u), k0(seed), k1(stage), block(0), have(0) {}
#else
  WAYNE_HD PhiloxStream(uint32_t seed, uint32_t stage, uint32_t c0_, uint32_t c2_,
                        uint32_t c3_)
      : c0(c0_), c2(c2_), c3(c3_), k0(seed), k1(stage), block(0), have(0) {}
#endif
  WAYNE_HD uint32_t next() {
    if (have == 0) {
      buf = philox4x32_10(c0, block, c2, c3, k0, k1);
      ++block;
      have = 4;
    }
    // words are consumed in order 0,1,2,3 (static indexing keeps buf in VGPRs)
    const uint32_t i = 4 - have;
    --have;
    return i == 0 ? buf.v[0] : i == 1 ? buf.v[1] : i == 2 ? buf.v[2] : buf.v[3];
  }
  WAYNE_HD void next2(uint32_t& a, uint32_t& b) { a = next(); b = next(); }   // two consecutive words
};

// xoshiro128+ seeded from one Philox block.
struct SeededStream {
  uint32_t s0, s1, s2, s3;
  WAYNE_HD SeededStream() : s0(0), s1(0), s2(0), s3(0) {}
  WAYNE_HD SeededStream(uint32_t seed, uint32_t stage, uint32_t c0, uint32_t c2, uint32_t c3, uint32_t c1 = 0u) {
#ifdef WAYNE_NEGCTL_ADDITIVE_KEY
    const u32x4 b = philox4x32_10(c0 + c2 + c3, c1, 0u, 0u, seed, stage);   // (see PhiloxStream)
#else
    const u32x4 b = philox4x32_10(c0, c1, c2, c3, seed, stage);
#endif
    s0 = b.v[0]; s1 = b.v[1]; s2 = b.v[2]; s3 = b.v[3];
  }
  WAYNE_HD uint32_t next() {
    const uint32_t result = s0 + s3;
    const uint32_t t = s1 << 9;
    s2 ^= s0;
    s3 ^= s1;
    s1 ^= s2;
    s0 ^= s3;
    s2 ^= t;
    s3 = (s3 << 11) | (s3 >> 21);
    return result;
  }
  // TWO words per state transition: a = s0 + s3 (the xoshiro128+ output) and b = s1 + s2 (the same
  // scrambler on the other two state words).  The map state -> (a, b) is balanced -- every 64-bit pair
  // has the same number of pre-images -- so pairs are uniform over the generator's period, and one pair
  // is what each consumer needs (angle + radius of a Box-Muller draw; table word + remainder word of a
  // sky draw; the U, V of a rejection trial): 9 VALU operations per pair instead of 16.
  WAYNE_HD void next2(uint32_t& a, uint32_t& b) {
    a = s0 + s3;
    b = s1 + s2;
    const uint32_t t = s1 << 9;
    s2 ^= s0;
    s3 ^= s1;
    s1 ^= s2;
    s0 ^= s3;
    s2 ^= t;
    s3 = (s3 << 11) | (s3 >> 21);
  }
};

// uint32 -> [1, 2) by writing 23 of its bits into a float's mantissa: the argument of v_sin_f32 /
// v_cos_f32 in REVOLUTIONS (period 1, so [1, 2) is as good as [0, 1)).  The LOW 23 bits, so that it is one
// instruction (v_and_or_b32; gfx950 has no shift-right-or): the weak lowest bits of a xoshiro+ word end up in the
// last mantissa bits of the angle, 2^-23 of a revolution.
WAYNE_HD float rev12(uint32_t x) {
  const uint32_t bits = (x & 0x7fffffu) | 0x3f800000u;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(bits);
#else
  float f;
  __builtin_memcpy(&f, &bits, 4);
  return f;
#endif
}

}  // namespace wayne
