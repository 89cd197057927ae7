// k_narrow: narrow PSF component as multinomials; k_lane: a bin's one-by-one electrons thrown by its own lane
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
// One lane per bin, 512 consecutive bins per workgroup, cells visited in
// lockstep with wave-level skipping (a cell is processed only while some lane
// still holds electrons).  Random words: the bin's STAGE_NARROW stream.
constexpr int kNarrowThreads = 512;
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
  const int k = blockIdx.x;                                    // (sub-sample fastest: see ThrowArgs::chunk_order)
  const int tid = threadIdx.x;
  const int w = (int)a.chunk_order[blockIdx.y] * kNarrowThreads + tid;
  const SubInfo si = a.sub[k];
  const int n0 = (w < a.W) ? a.nsplit[(size_t)k * a.W + w] : 0;   // narrow electrons of the bin's multinomial
  if (!__syncthreads_or(n0 > 0)) return;

  float x = 0.f, y = 0.f, sg = 1.f;
  int ic0 = 0, jc0 = 0;
  if (n0 > 0) {
    x = (float)a.xpos[(size_t)k * a.W + w];
    y = (float)a.ypos[(size_t)k * a.W + w];
    sg = (float)a.sigl[w];
    ic0 = (int)floorf(x);
    jc0 = (int)floorf(y);
  }
  // workgroup tile = bounding box of its bins' windows, clipped to [1, N)
  if (tid == 0) { s_box[0] = 0x7FFFFFFF; s_box[1] = -0x7FFFFFFF; s_box[2] = 0x7FFFFFFF; s_box[3] = -0x7FFFFFFF; }
  __syncthreads();
  if (n0 > 0) {
    atomicMin(&s_box[0], ic0 - kNarrowR); atomicMax(&s_box[1], ic0 + kNarrowR + 1);
    atomicMin(&s_box[2], jc0 - kNarrowR); atomicMax(&s_box[3], jc0 + kNarrowR + 1);
  }
  __syncthreads();
  int tx0 = max(s_box[0], 1), tx1 = min(s_box[1], a.N), ty0 = max(s_box[2], 1), ty1 = min(s_box[3], a.N);
  int tw = max(tx1 - tx0, 0), th = max(ty1 - ty0, 0);
  if ((long long)tw * th > kNarrowTile) { th = min(th, kNarrowTile / max(tw, 1)); if (th < 1) { th = 0; tw = 0; } }
  const int tarea = tw * th;
  for (int i = tid; i < tarea; i += kNarrowThreads) tile[i] = 0;

  // (waves without a multinomial bin skip the work altogether)
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
// k_lane : a bin's one-by-one electrons, thrown by the bin's own lane
// ---------------------------------------------------------------------------
// In split mode what is left to throw electron by electron is, per bin, a few hundred wide-PSF
// electrons (sigma_h ~ 5-6 px: too spread out for a multinomial to pay) or the whole of a thinly
// populated bin.  Sharing those out over lanes by electron number (k_throw) costs a prefix search per
// workgroup and a bin change every few hundred electrons somewhere in every wave; here lane = bin, as in
// k_narrow: position and sigma are loop constants, neighbouring bins hold nearly the same number of
// electrons (the spectrum is smooth), and the per-electron work is the draw and the deposit alone.
// Electron j of the bin takes pair j of the bin's STAGE_LANE stream; the first n_wide electrons take
// sigma_h (pyparallel_menu.c:89-107).  Same arithmetic as k_throw's Philox mode.
constexpr int kLaneThreads = 512;
// The tile must hold practically every electron: one that falls outside takes the global-atomic path INSIDE the
// loop (four flat-plane loads, the fp64 flat polynomial, a 64-bit atomic: microseconds of latency with the other
// 63 lanes of the wave idle).  At a margin of 22 px (3.7 sigma_h) 3 % of the wave-iterations had such a lane and
// the kernel ran at 220 cycles per iteration instead of ~135; 30 px is 5 sigma_h.
constexpr int kLaneMargin = 30;
constexpr int kLaneTile = 5376;         // ints of LDS (21 KB): 512 bins span ~20 px of the trace, + 2 x margin, by 2 x margin + a few rows

template <int FLUSH>
__global__ __launch_bounds__(kLaneThreads) void k_lane(ThrowArgs a) {
  __shared__ int tile[kLaneTile];
  __shared__ int s_box[4];
  const int k = blockIdx.x;                                    // (sub-sample fastest: see ThrowArgs::chunk_order)
  const int tid = threadIdx.x;
  const int w = (int)a.lane_order[blockIdx.y] * kLaneThreads + tid;
  const size_t kw = (size_t)k * a.W + (w < a.W ? w : 0);
  const int n = (w < a.W) ? a.nlane[kw] : 0;
  if (!__syncthreads_or(n > 0)) return;
  const SubInfo si = a.sub[k];

  float x = -1e30f, y = -1e30f, ch = 0.f, cl = 0.f;
  int nw = 0;
  if (n > 0) {
    x = (float)a.xpos[kw];
    y = (float)a.ypos[kw];
    const float sh = (float)a.sigh[w], sl = (float)a.sigl[w];
    ch = (-1.3862943611198906f * sh) * sh;
    cl = (-1.3862943611198906f * sl) * sl;
    nw = min(max(a.nwide[kw], 0), n);
  }
  // workgroup tile: bounding box of its bins' positions +- margin, clipped to [1, N) and to the LDS budget
  if (tid == 0) { s_box[0] = 0x7FFFFFFF; s_box[1] = -0x7FFFFFFF; s_box[2] = 0x7FFFFFFF; s_box[3] = -0x7FFFFFFF; }
  __syncthreads();
  if (n > 0 && fabsf(x) < 1e6f && fabsf(y) < 1e6f) {
    const int ic = (int)floorf(x), jc = (int)floorf(y);
    atomicMin(&s_box[0], ic - kLaneMargin); atomicMax(&s_box[1], ic + kLaneMargin + 1);
    atomicMin(&s_box[2], jc - kLaneMargin); atomicMax(&s_box[3], jc + kLaneMargin + 1);
  }
  __syncthreads();
  int tx0 = max(s_box[0], 1), tx1 = min(s_box[1], a.N), ty0 = max(s_box[2], 1), ty1 = min(s_box[3], a.N);
  int tw = max(tx1 - tx0, 0), th = max(ty1 - ty0, 0);
  while ((long long)tw * th > kLaneTile && th > 1) { ty0 += 1; th = max(th - 2, 1); }
  while ((long long)tw * th > kLaneTile && tw > 1) { tx0 += 1; tw = max(tw - 2, 1); }
  if ((long long)tw * th > kLaneTile) { tw = 0; th = 0; }
  const int tarea = tw * th;
  for (int i = tid; i < tarea; i += kLaneThreads) tile[i] = 0;
  __syncthreads();

  // Electron j of the bin takes pair j of the bin's stream; the first nw take sigma_h.  A wave whose bins all
  // had their narrow electrons taken by k_narrow (nw = n everywhere: the usual case) runs with sigma as a loop
  // constant: first the iterations EVERY lane has, with no per-lane test at all (neighbouring bins hold nearly the
  // same number of electrons), then the tail, where the stream still advances in every lane and only the deposit
  // is suppressed (position far off the frame) for the lanes that are done.  A wave with thin, unsplit bins
  // selects sigma per electron.
  const int tw4 = tw * 4;
  auto throw_one = [&](SeededStream& rng, float c, float px, float py) {
    uint32_t wa, wb;
    rng.next2(wa, wb);
    const float rev = rev12(wa);
    const float Rs = __builtin_amdgcn_sqrtf(c * __builtin_amdgcn_logf(u01f(wb)));
    const int xi = (int)fmaf(__builtin_amdgcn_cosf(rev), Rs, px);   // C truncation toward zero (:91-92)
    const int yi = (int)fmaf(__builtin_amdgcn_sinf(rev), Rs, py);
    const int lx = xi - tx0, ly = yi - ty0;
    if ((unsigned)lx < (unsigned)tw && (unsigned)ly < (unsigned)th)
      atomicAdd((int*)((char*)tile + (__umul24(ly, tw4) + (lx << 2))), 1);
    else if (xi > 0 && xi < a.N && yi > 0 && yi < a.N)               // (:93)
      deposit_global<FLUSH>(a, si, xi, yi, 1);
  };
  int cmin = n, cmax = n;
  for (int off = 32; off > 0; off >>= 1) {
    cmin = min(cmin, __shfl_xor(cmin, off));
    cmax = max(cmax, __shfl_xor(cmax, off));
  }
  cmin = __builtin_amdgcn_readfirstlane(cmin);
  cmax = __builtin_amdgcn_readfirstlane(cmax);
  if (cmax > 0) {
    SeededStream rng(a.seed, STAGE_LANE, (uint32_t)w, (uint32_t)k + a.subsample0, a.exposure);
    if (!__any(n > nw)) {
      for (int j = 0; j < cmin; ++j) throw_one(rng, ch, x, y);
      for (int j = cmin; j < cmax; ++j) {
        const bool live = j < n;
        throw_one(rng, ch, live ? x : -1e30f, live ? y : -1e30f);
      }
    } else {
      for (int j = 0; j < cmax; ++j) {
        const bool live = j < n;
        throw_one(rng, (j < nw) ? ch : cl, live ? x : -1e30f, live ? y : -1e30f);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < tarea; i += kLaneThreads) {
    const int m = tile[i];
    if (m > 0) {
      const int ly = i / tw, lx = i - ly * tw;
      deposit_global<FLUSH>(a, si, tx0 + lx, ty0 + ly, m);
    }
  }
}

}  // namespace wayne
