// k_narrow: narrow PSF component as multinomials, sparse bins lane per bin
#pragma once
#include "common.h"
#include "k_throw.h"

namespace wayne {

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
  __shared__ float s_q[kNarrowCells][kNarrowThreads];   // CONDITIONAL row probabilities of each lane's bin (see below)
  __shared__ int s_box[4];
  __shared__ float s_fc[10];                            // stirling_tail(0..9), indexed per lane in the rejection sampler
  if (threadIdx.x < 10) s_fc[threadIdx.x] = (float)kStirlingSmall[threadIdx.x];
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
    // rows, centre-out (c = 0 centre, odd c -> +((c+1)/2), even c -> -(c/2)): what the chain needs of row c is the
    // probability of landing in it GIVEN that none of the rows before it was hit, mass_c / (mass not yet visited) --
    // the same for every column of the bin (x and y are independent), so it is computed once, here, with the
    // not-yet-visited mass taken as the sum of the two remaining tails (a running 1 - sum would lose the far rows
    // to cancellation)
    {
      const float f = y - (float)jc0;
      float up = upper_tail((1.f - f) * inv_s), lo = upper_tail(f * inv_s);   // mass above / below the centre row
      s_q[0][tid] = 1.f - up - lo;
      for (int c = 1; c < kNarrowCells; ++c) {
        const int d = (c + 1) >> 1;
        const float rem = up + lo;
        float Q;
        if (c & 1) { const float nx = upper_tail(((float)(d + 1) - f) * inv_s); Q = up - nx; up = nx; }
        else       { const float nx = upper_tail(((float)d + f) * inv_s);       Q = lo - nx; lo = nx; }
        s_q[c][tid] = fminf(fmaxf(M::div_(Q, rem), 0.f), 1.f);
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
        n_col = binomial<M>(n_rem, pc, rng, s_fc);
        n_rem -= n_col;
      }
      if (!__any(n_col > 0.f)) continue;
      // rows of this column
      float m_rem = n_col;
      for (int r = 0; r < kNarrowCells; ++r) {
        if (!__any(m_rem > 0.f)) break;
        float m = 0.f;
        if (m_rem > 0.f) {
          m = binomial<M>(m_rem, s_q[r][tid], rng, s_fc);
          m_rem -= m;
        }
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

}  // namespace wayne
