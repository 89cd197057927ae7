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
//
// POOLED ROWS.  The row chains are two thirds of the draws, and the 16 bins of a
// DPP row (w >> 4: "group") sit at practically the same height with practically
// the same sigma_l, so their row distributions q_b(t) over a common window of 14
// absolute rows nearly coincide.  Write
//     q_b = Z (qbar / Z) + (1 - Z) r_b,   qbar(t) = min_b q_b(t),  Z = sum_t qbar(t),
//     r_b = (q_b - qbar) / (1 - Z):
// an electron takes its row from the distribution the whole group shares with
// probability Z and from its bin's own residual otherwise -- exact for every
// Z in (0, 1], Z = 0.99+ on a spectrum.  Rows and columns are independent, so a
// bin (1) thins its n electrons into Binomial(n, Z) "common" and the rest
// "residual", (2) runs its column chain on the common ones and adds the column
// counts into the group's 16 per-column totals (LDS), (3) throws the residual
// handful one by one (column from the gaussian itself, row by inverse CDF on
// q_b - qbar).  Then (4) lane j of the group draws the rows of column j's total
// in ONE chain (stream STAGE_POOL): 16 row chains per group instead of ~6 per
// bin.  A group pools when it has >= 2 multinomial bins within 2 columns,
// 0.25 sigma in y and 10 % in sigma of one another (Z >~ 0.8) and <= 2^24
// electrons; any other group runs a column and a row chain per bin as above.
// oracle/split_oracle.c (so_group_pools, so_narrow_pooled) is the same procedure.
// (kNarrowThreads = 512 bins per workgroup: plan_consts.h)
constexpr int kNarrowCells = 2 * kNarrowR + 1;
constexpr int kPoolRows = 2 * kNarrowR + 2;   // rows of a group's common window: jc_min - R .. jc_min + R + 1
constexpr int kNarrowTile = 1536;       // ints of LDS for the workgroup's tile (its bins span ~15 x 1 px + the 13 x 13 windows)

// P(Z > t) of the standard normal for t >= 0 = erfc(t / sqrt 2) / 2, by the Chebyshev fit of Numerical Recipes
// (erfcc: t' exp(-z^2 + poly(t')), t' = 1 / (1 + z/2); fractional error < 1.2e-7 in exact arithmetic, < 5e-6 here
// out to 6.5 sigma, absolute error < 2e-7): a third of the instructions of the library erfcf, which is what a bin's
// ~25 cell edges cost.  Taken as 0 beyond 6.5 sigma (4e-11: a 2^24-electron bin would put 7e-4 electrons there);
// skipped altogether when the whole wave is out there, as the outer cells of the window are.
constexpr float kTailCut = 6.5f;
template <class M>
__device__ __forceinline__ float upper_tail(float t) {
  if (__all(t > kTailCut)) return 0.f;
  const float z = t * 0.70710678118654752f;
  const float u = M::div_(1.f, 1.f + 0.5f * z);
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
  return (t > kTailCut) ? 0.f : (0.5f * u) * M::exp_(p);
}

template <int FLUSH, bool FAST>
__global__ __launch_bounds__(kNarrowThreads) void k_narrow(ThrowArgs a) {
  typedef typename std::conditional<FAST, FastMath, ExactMath<float> >::type M;
  __shared__ int tile[kNarrowTile];
  // per lane: the running sums of a pooling bin's residual row masses q_b(t) - qbar(t), t = 0..13; any other bin's CONDITIONAL row
  // probabilities in visiting order (see below)
  __shared__ float s_q[kPoolRows][kNarrowThreads];
  __shared__ int s_pool[kNarrowThreads];                // [group][column of the group's window]: pooled column counts
  __shared__ float s_cond[kNarrowThreads];              // [group][i]: conditional probability of the i-th row visited by a pooled chain
  __shared__ int s_box[4];
  __shared__ float s_fc[10];                            // stirling_tail(0..9), indexed per lane in the rejection sampler
  if (threadIdx.x < 10) s_fc[threadIdx.x] = (float)kStirlingSmall[threadIdx.x];
  const int k = blockIdx.x;                                    // (sub-sample fastest: see ThrowArgs::chunk_order)
  const int tid = threadIdx.x;
  // (the workgroup's chunk by a scalar load of the argument word that holds its byte; the bin's count, position and
  // sigma asked for together, whether or not the count turns out positive: see k_lane)
  const uint32_t order_word = reinterpret_cast<const uint32_t*>(a.chunk_order)[blockIdx.y >> 2];
  const int w = (int)((order_word >> (8 * (blockIdx.y & 3))) & 0xFFu) * kNarrowThreads + tid;
  const SubInfo si = a.sub[k];
  int n0 = 0;                                                  // narrow electrons of the bin's multinomial
  double xd = 0., yd = 0., sgd = 1.;
  if (w < a.W) {
    const size_t kw = (size_t)k * a.W + w;
    n0 = a.nsplit[kw];
    xd = a.xpos[kw];
    yd = a.ypos[kw];
    sgd = a.sigl[w];
  }
  if (!__syncthreads_or(n0 > 0)) return;
  // the bin's position as pixel (ic0, jc0) + fraction (fx, fy) of it, split in fp64 (common.h, bin_local).  A bin whose
  // position is not sane keeps its place in the group's count (so_group_pools) and draws nothing: within the window of
  // +-6 px none of its electrons can reach the frame
  const bool split = n0 > 0;
  BinLocal bl{};
  if (split) bl = bin_local(xd, yd);
  const bool act = split && bl.sane;
  const float sg = split ? (float)sgd : 1.f;
  const int ic0 = bl.ox, jc0 = bl.oy;
  s_pool[tid] = 0;
  // workgroup tile = bounding box of its bins' windows, clipped to [1, N)
  if (tid == 0) { s_box[0] = 0x7FFFFFFF; s_box[1] = -0x7FFFFFFF; s_box[2] = 0x7FFFFFFF; s_box[3] = -0x7FFFFFFF; }
  __syncthreads();
  {
    // (reduced over the wave first: 512 lanes hammering four LDS words with atomics serialise for microseconds)
    const int x_lo = wave_mini(act ? ic0 - kNarrowR : 0x7FFFFFFF), x_hi = wave_maxi(act ? ic0 + kNarrowR + 1 : -0x7FFFFFFF);
    const int y_lo = wave_mini(act ? jc0 - kNarrowR : 0x7FFFFFFF), y_hi = wave_maxi(act ? jc0 + kNarrowR + 1 : -0x7FFFFFFF);
    if ((tid & 63) == 0) {
      atomicMin(&s_box[0], x_lo); atomicMax(&s_box[1], x_hi);
      atomicMin(&s_box[2], y_lo); atomicMax(&s_box[3], y_hi);
    }
  }
  __syncthreads();
  int tx0 = max(s_box[0], 1), tx1 = min(s_box[1], a.N), ty0 = max(s_box[2], 1), ty1 = min(s_box[3], a.N);
  int tw = max(tx1 - tx0, 0), th = max(ty1 - ty0, 0);
  if ((long long)tw * th > kNarrowTile) { th = min(th, kNarrowTile / max(tw, 1)); if (th < 1) { th = 0; tw = 0; } }
  const int tarea = tw * th;
  for (int i = tid; i < tarea; i += kNarrowThreads) tile[i] = 0;

  auto deposit = [&](int ci, int rj, int m) {
    const int lx = ci - tx0, ly = rj - ty0;
    if ((unsigned)lx < (unsigned)tw && (unsigned)ly < (unsigned)th)
      atomicAdd(&tile[ly * tw + lx], m);
    else if (ci > 0 && ci < a.N && rj > 0 && rj < a.N)       // (:93)
      deposit_global<FLUSH>(a, si, ci, rj, m);
  };

  // (waves without a multinomial bin skip the work altogether)
  const bool any_multi = __any(act);
  const float inv_s = 1.f / sg;
  const int grp = tid & ~15, gl = tid & 15;
  bool pool = false;
  int X0 = 0, J0 = 0, res = 0;
  float Z = 1.f, Rb = 0.f, yl = 0.f;
  if (any_multi) {
    // does the lane's group pool its rows?  (so_group_pools)
    const int cnt = row16_sum(split ? 1 : 0), cnt_sane = row16_sum(act ? 1 : 0), tot = row16_sum(split ? n0 : 0);
    const int icmin = row16_mini(act ? ic0 : 0x7FFFFFFF), icmax = row16_maxi(act ? ic0 : -0x7FFFFFFF);
    const int jcmin = row16_mini(act ? jc0 : 0x7FFFFFFF);
    X0 = icmin - kNarrowR;
    J0 = jcmin - kNarrowR;
    // the bin's height above the bottom row of the group's window (6 .. 8 px when the group pools): one rounding of fp64
    yl = act ? (float)(yd - (double)J0) : 0.f;
    const float ymin = row16_min(act ? yl : 3e38f), ymax = row16_max(act ? yl : -3e38f);
    const float smin = row16_min(split ? sg : 3e38f), smax = row16_max(split ? sg : 0.f);
    pool = cnt >= 2 && cnt_sane == cnt && icmax - icmin <= 2 && (ymax - ymin) <= 0.25f * smin && smax <= 1.1f * smin &&
           tot <= 16777216;

    float qh[kPoolRows];
#pragma unroll
    for (int t = 0; t < kPoolRows; ++t) qh[t] = 3e38f;
    if (act && pool) {
      // the bin's masses on the rows [J0 + t, J0 + t + 1): differences of two tails on the same side of y
      float Ep = -yl, Ap = upper_tail<M>(fabsf(Ep) * inv_s), S = 0.f;
#pragma unroll
      for (int t = 0; t < kPoolRows; ++t) {
        const float E = (float)(t + 1) - yl, A = upper_tail<M>(fabsf(E) * inv_s);
        const float m = (Ep >= 0.f) ? Ap - A : ((E <= 0.f) ? A - Ap : 1.f - Ap - A);
        qh[t] = fmaxf(m, 0.f);
        S += qh[t];
        Ep = E; Ap = A;
      }
#pragma unroll
      for (int t = 0; t < kPoolRows; ++t) qh[t] = M::div_(qh[t], S);
    } else if (act) {
      // rows, centre-out (c = 0 centre, odd c -> +((c+1)/2), even c -> -(c/2)): what the chain needs of row c is the
      // probability of landing in it GIVEN that none of the rows before it was hit, mass_c / (mass not yet visited) --
      // the same for every column of the bin (x and y are independent), so it is computed once, here, with the
      // not-yet-visited mass taken as the sum of the two remaining tails (a running 1 - sum would lose the far rows
      // to cancellation)
      const float f = bl.fy;
      float up = upper_tail<M>((1.f - f) * inv_s), lo = upper_tail<M>(f * inv_s);   // mass above / below the centre row
      s_q[0][tid] = 1.f - up - lo;
      for (int c = 1; c < kNarrowCells; ++c) {
        const int d = (c + 1) >> 1;
        const float rem = up + lo;
        float Q;
        if (c & 1) { const float nx = upper_tail<M>(((float)(d + 1) - f) * inv_s); Q = up - nx; up = nx; }
        else       { const float nx = upper_tail<M>(((float)d + f) * inv_s);       Q = lo - nx; lo = nx; }
        s_q[c][tid] = fminf(fmaxf(M::div_(Q, rem), 0.f), 1.f);
      }
    }
    if (__any(pool)) {
      // what the group's bins share: qbar(t) = min over the bins; its tails about the centre row, summed inwards
      float qb[kPoolRows];
#pragma unroll
      for (int t = 0; t < kPoolRows; ++t) qb[t] = row16_min(qh[t]);
      float PL[kNarrowR + 1], SU[kPoolRows + 1];
      PL[0] = qb[0];
#pragma unroll
      for (int t = 1; t <= kNarrowR; ++t) PL[t] = PL[t - 1] + qb[t];
      SU[kPoolRows] = 0.f;
#pragma unroll
      for (int t = kPoolRows - 1; t > kNarrowR; --t) SU[t] = SU[t + 1] + qb[t];
      Z = fminf(PL[kNarrowR] + SU[kNarrowR + 1], 1.f);
      if (act && pool) {
#pragma unroll
        for (int t = 0; t < kPoolRows; ++t) {
          Rb += qh[t] - qb[t];
          s_q[t][tid] = Rb;                     // running sum: the residual rows' inverse CDF is a count of thresholds
        }
      }
      if (pool && gl == 0) {
        // rows in visiting order: centre (6), then 7, 5, 8, 4, ...: mass over the two tails not yet visited
        s_cond[grp] = fminf(fmaxf(M::div_(qb[kNarrowR], PL[kNarrowR] + SU[kNarrowR + 1]), 0.f), 1.f);
#pragma unroll
        for (int i = 1; i < kPoolRows; ++i) {
          const int u = (i + 1) >> 1;
          float rem, q;
          if (i & 1) { q = qb[kNarrowR + u]; rem = SU[kNarrowR + u] + (u <= kNarrowR ? PL[u <= kNarrowR ? kNarrowR - u : 0] : 0.f); }
          else       { q = qb[kNarrowR - u]; rem = SU[kNarrowR + 1 + u] + PL[kNarrowR - u]; }
          s_cond[grp + i] = fminf(fmaxf(M::div_(q, rem), 0.f), 1.f);
        }
      }
    }
  }
  __syncthreads();

  if (any_multi) {
    SeededStream rng(a.seed, STAGE_NARROW, (uint32_t)w, (uint32_t)k + a.subsample0, a.exposure);
    const float fx = bl.fx;
    float up = upper_tail<M>((1.f - fx) * inv_s), lo = upper_tail<M>(fx * inv_s);
    float n_rem = act ? (float)n0 : 0.f;
    // c = -1: a pooling bin thins its electrons (common ~ Binomial(n, Z)); c >= 0: the column chain
    for (int c = __any(pool) ? -1 : 0; c < kNarrowCells; ++c) {
      if (!__any(n_rem > 0.f)) break;
      // this column's mass and the mass of everything not yet visited (before it)
      const int d = (c + 1) >> 1;
      const int ci = (c <= 0) ? ic0 : ((c & 1) ? ic0 + d : ic0 - d);
      float P, rem;
      if (c < 0) { P = Z; rem = 1.f; }
      else if (c == 0) { P = 1.f - up - lo; rem = 1.f; }
      else if (c & 1) { const float nx = upper_tail<M>(((float)(d + 1) - fx) * inv_s); P = up - nx; rem = up + lo; up = nx; }
      else            { const float nx = upper_tail<M>(((float)d + fx) * inv_s);       P = lo - nx; rem = up + lo; lo = nx; }
      float n_col = 0.f;
      if (n_rem > 0.f && (c >= 0 || pool)) {
        const float pc = (c < 0) ? P : fminf(fmaxf(M::div_(P, rem), 0.f), 1.f);
#ifdef WAYNE_TIMING_COLPOOL
        // TIMING BUILD (wrong frames; HISTORY.md section 9, "column chains pooled over 4 bins"): what the kernel would cost
        // if three of four column chains did not exist -- the chains of waves 2..7 are replaced by a rounding (whole
        // waves, so the issue slots really go), their thinning draw and everything downstream stays
        if (c >= 0 && (tid >> 6) >= 2) n_col = fminf(floorf(fmaf(n_rem, pc, 0.5f)), n_rem);
        else
#endif
        n_col = binomial<M>(n_rem, pc, rng, s_fc);
        if (c < 0) { res = (int)(n_rem - n_col); n_rem = n_col; n_col = 0.f; }
        else n_rem -= n_col;
      }
      if (pool) {
        if (n_col > 0.f) atomicAdd(&s_pool[grp + (ci - X0)], (int)n_col);
        n_col = 0.f;
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
          deposit(ci, (r == 0) ? jc0 : ((r & 1) ? jc0 + e : jc0 - e), (int)m);
        }
      }
    }
    // the residual electrons of the pooling bins, one by one: two pairs each
    if (__any(res > 0)) {
      const float cs = (-1.3862943611198906f * sg) * sg;
      for (int e = 0; __any(e < res); ++e) {
        if (e < res) {
          uint32_t wa, wb, va, vb;
          rng.next2(wa, wb);
          rng.next2(va, vb);
          const float Rs = __builtin_amdgcn_sqrtf(cs * __builtin_amdgcn_logf(u01f(wb)));
          const int col = ic0 + min(max(floor_to_int(fmaf(__builtin_amdgcn_cosf(rev12(wa)), Rs, fx)), -kNarrowR), kNarrowR);
          const float u = u01f(va) * Rb;
          int row = 0;
#pragma unroll
          for (int t = 0; t < kPoolRows - 1; ++t) row += (u >= s_q[t][tid]) ? 1 : 0;
          deposit(col, J0 + row, 1);
        }
      }
    }
    // the pooled columns' rows: lane j of a group takes column X0 + j
    if (__any(pool)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float left = pool ? (float)s_pool[tid] : 0.f;
      if (__any(left > 0.f)) {
        SeededStream rp(a.seed, STAGE_POOL, (uint32_t)w >> 4, (uint32_t)k + a.subsample0, a.exposure, (uint32_t)gl);
        for (int i = 0; i < kPoolRows; ++i) {
          if (!__any(left > 0.f)) break;
          float m = 0.f;
          if (left > 0.f) {
            m = binomial<M>(left, s_cond[grp + i], rp, s_fc);
            left -= m;
          }
          if (m > 0.f) {
            const int u = (i + 1) >> 1;
            deposit(X0 + gl, J0 + ((i & 1) ? kNarrowR + u : kNarrowR - u), (int)m);
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
// (kLaneThreads = 512 bins per workgroup: plan_consts.h)
// The tile must hold practically every electron: one that falls outside takes the global-atomic path INSIDE the
// loop (four flat-plane loads, the fp64 flat polynomial, a 64-bit atomic: microseconds of latency with the other
// 63 lanes of the wave idle).  At a margin of 22 px (3.7 sigma_h) 3 % of the wave-iterations had such a lane and
// the kernel ran at 220 cycles per iteration instead of ~135; 30 px is 5 sigma_h.
//
// Better still, the tile can hold EVERY electron: the radius' uniform is >= 2^-34 (see the electron loop), so an
// electron lands within sigma sqrt(2 ln 2^34) = 6.87 sigma of its bin.  A workgroup whose tile -- the bins' bounding
// box +- (6.9 sigma_max + 1) px -- fits the LDS budget and lies inside the frame needs no bounds test at all: the deposit is
// cvt, cvt, lshl_add, mad, ds_add (the tile origin folded into one scalar), four vector instructions fewer per
// electron and no branch in the loop.  Other workgroups (frame edge, huge sigma, wild positions) keep the test and
// the fixed margin.
constexpr int kLaneMargin = 30;
constexpr int kLaneReachMax = 48;       // largest margin of a test-free tile (sigma_h up to 6.8 px)
constexpr int kLaneTile = 9216;         // ints of LDS (36 KB): 512 bins span ~20 px of the trace, + 2 x margin, by 2 x margin + a few rows

// BATCHES.  A finely sampled scan (the reference's default 10 ms sampling: K = 2233 sub-samples of ~2.5 electrons
// per bin) launches K x chunks workgroups of ~1300 electrons each.  A workgroup can take `kb` consecutive sub-samples
// of its chunk (the host's choice, 1 on a coarsely sampled exposure): one tile geometry for the batch (the union of
// its sub-samples' bins), the tile flushed -- with that sub-sample's flat -- and left clean after every sub-sample.
// THIN (the host expects few electrons per workgroup and sub-sample): an electron that is the first on its tile cell
// (the LDS atomic returns 0) puts the cell on a list, and the flush then visits the list's ~1000 cells instead of
// scanning the tile's 9216 with a tenth of the lanes finding anything.  Same electrons, same streams, same sums
// (integer accumulation commutes): frames do not depend on kb or THIN.  Measured on cfg1 (K = 2233): kb = 9 + THIN
// takes k_lane from 0.279 to 0.230 ms; the same batching of k_prep_sub and k_narrow was built and measured SLOWER
// (0.259 -> 0.286, 0.039 -> 0.057 ms: more registers, a barrier per sub-sample) and is not kept -- those kernels'
// time on such an exposure is arithmetic per (bin, sub-sample): ~2.2 Philox blocks and an fp64 product-of-uniforms
// search per stellar Poisson draw, not prologues (profiles/r03/cfg1_batches.txt).
// (kLaneListCap = 4096 cells on a THIN flush list, kLaneBatchMax = 32 sub-samples per workgroup: plan_consts.h)

// FUSED (thin exposures, split mode, no k_throw and no k_narrow launched): there is no k_prep_sub launch either -- the
// lane works out its bin's position, count and routing itself, per sub-sample, with k_prep_sub's own code (plan_bin),
// and throws at once.  A finely sampled scan is 10^7 (bin, sub-sample) pairs of ~2.5 electrons: written by k_prep_sub
// and read back here, the six K x W arrays of intermediates were ~900 MB of traffic per exposure and the two kernels
// spent two thirds of their wave-cycles waiting on them.  Only the counts are still stored (for
// wayne_exposure_debug_fetch); positions are worked out again by a k_prep_sub launch if a caller asks for them.
template <int FLUSH, bool THIN, bool BATCH, bool FUSED>
__device__ __forceinline__ void lane_body(const ThrowArgs& a, const PrepArgs& p, const CosmicArgs& ca) {
  __shared__ int tile[kLaneTile];
  __shared__ SubInfo s_sub[FUSED ? kLaneBatchMax : 1];
  __shared__ uint32_t s_hits;
  __shared__ unsigned short s_list[THIN ? kLaneListCap : 1];
  __shared__ int s_cnt[2];                // THIN: cells on the flush list; two counters used in turn (see the flush)
  __shared__ int s_box[4];
  __shared__ int s_reach;                 // max over the lanes of 6.9 sigma + 1 (float bits; 0x7F800000 if a lane is not sane)
  // (sub-sample fastest: see ThrowArgs::chunk_order; BATCH = false: one sub-sample, as k_narrow)
  const int k0 = BATCH ? (int)blockIdx.x * a.kb : (int)blockIdx.x, k1 = BATCH ? min(k0 + a.kb, a.K) : k0 + 1;
  const int tid = threadIdx.x;
  // (the chunk of this workgroup from the kernel arguments by a scalar load of the word that holds its byte: indexed as
  // a byte array it is a vector load from the argument segment, a memory latency in front of every other load here)
  const uint32_t order_word = reinterpret_cast<const uint32_t*>(a.lane_order)[blockIdx.y >> 2];
  const int w = (int)((order_word >> (8 * (blockIdx.y & 3))) & 0xFFu) * kLaneThreads + tid;
  const bool inw = w < a.W;
  // One sub-sample per workgroup (the rule: BATCH = false): everything the bin needs is asked for at once -- its
  // sigmas, its count, position and wide count -- instead of count first, then (if positive) the rest, then all of it
  // again in the throw loop: four memory latencies in a row per workgroup of a kernel whose launch-independent cost
  // was 30 us (scripts/thrower_vs_electrons.py)
  int n_one = 0, nw_one = 0;
  BinLocal bl_one{};                      // the bin's position as pixel + fraction (common.h, bin_local)
  float ch = 0.f, cl = 0.f, sh = 0.f, sl = 0.f;
  if (inw) {
    double sh_d = a.sigh[w], sl_d = a.sigl[w];
    if (!FUSED && !BATCH) {
      const size_t kw0 = (size_t)k0 * a.W + w;
      n_one = a.nlane[kw0];
      const double xd = a.xpos[kw0], yd = a.ypos[kw0];
      nw_one = a.nwide[kw0];
      bl_one = bin_local(xd, yd);
      nw_one = min(max(nw_one, 0), n_one);
    }
    sh = (float)sh_d; sl = (float)sl_d;
    ch = (-1.3862943611198906f * sh) * sh;
    cl = (-1.3862943611198906f * sl) * sl;
  }
  // A sigma that is not finite (or whose square is not): the reference's (int) of a non-finite position keeps none of
  // the electrons that take it (pyparallel_menu.c:91-93).  Settled here, once per bin -- such electrons are thrown at
  // -1e30 -- so that every sum in the electron loop is finite and its floor needs no guard.
  const bool bad_h = !(ch > -3e38f), bad_l = !(cl > -3e38f);
  if (bad_h) ch = 0.f;
  if (bad_l) cl = 0.f;
  // FUSED: the bin's per-wavelength inputs of the counts chain, the batch's SubInfo records, and the cosmic-ray hits
  // that k_prep_sub's workgroups add on their way in
  double f_wl = 0., f_flux = 0., f_sens = 0., f_dlam = 0., f_ratio = 0., f_sigl = 0.;
  unsigned long long f_electrons = 0ull;
  if (FUSED) {
    if (ca.rate >= 0.) {
      const int n_wg = gridDim.x * gridDim.y;
      for (int r = blockIdx.y * gridDim.x + blockIdx.x; r < ca.R; r += n_wg) { cosmic_hits(ca, r, &s_hits); __syncthreads(); }
    }
    if (inw) { f_wl = p.wl[w]; f_flux = p.flux[w]; f_sens = p.wa.sens[w]; f_dlam = p.wa.dlam[w]; f_ratio = p.wa.ratio[w]; f_sigl = p.wa.sigl[w]; }
    if (tid < k1 - k0) {
      const int k = k0 + tid;
      s_sub[tid] = make_sub_info(p, k, p.x_ref[k], p.y_ref[k], p.tr + kTrStride * (size_t)k, 0u, 1e300, -1e300, 1e300, -1e300);
    }
  }
  // the batch's populated bins: bounding box and reach
  int x_lo = 0x7FFFFFFF, x_hi = -0x7FFFFFFF, y_lo = 0x7FFFFFFF, y_hi = -0x7FFFFFFF;
  float reach = 0.f;
  bool any = false;
  for (int k = k0; k < k1; ++k) {
    const size_t kw = (size_t)k * a.W + (inw ? w : 0);
    int n = 0, nw = 0;
    int ic = 0, jc = 0;
    bool sane = false;
    if (FUSED) {
      // every bin of the chunk, populated or not, and both sigmas (the counts are not drawn twice for a box); the
      // positions without their division, good to 1e-10 px, and the box a pixel larger all round
      if (inw) {
        float x = 0.f, y = 0.f;
        bin_position_bound(p, f_wl, p.tr + kTrStride * (size_t)k, p.x_ref[k], p.y_ref[k], &x, &y);
        sane = fabsf(x) < 1e6f && fabsf(y) < 1e6f;
        ic = (int)floorf(x); jc = (int)floorf(y);
        n = 2; nw = 1;
      }
    } else if (!BATCH) {
      n = n_one;
      if (n > 0) { sane = bl_one.sane; ic = bl_one.ox; jc = bl_one.oy; nw = nw_one; }
    } else if (inw) {
      n = a.nlane[kw];
      if (n > 0) {
        const BinLocal b = bin_local(a.xpos[kw], a.ypos[kw]);
        sane = b.sane; ic = b.ox; jc = b.oy;
        nw = min(max(a.nwide[kw], 0), n);
      }
    }
    if (n > 0) {
      any = true;
      // (each sigma the bin uses on its own: fmaxf would step over a NaN)
      const bool sig_ok = (nw > 0 ? (sh >= 0.f && sh < 1e6f) : true) && (n > nw ? (sl >= 0.f && sl < 1e6f) : true);
      const float smax = fmaxf(nw > 0 ? sh : 0.f, n > nw ? sl : 0.f);
      float r = sig_ok ? 6.9f * smax + 1.f : __int_as_float(0x7F800000);
      if (sane) {
        constexpr int spare = FUSED ? 1 : 0;
        x_lo = min(x_lo, ic - spare); x_hi = max(x_hi, ic + spare); y_lo = min(y_lo, jc - spare); y_hi = max(y_hi, jc + spare);
      } else {
        r = __int_as_float(0x7F800000);
      }
      reach = fmaxf(reach, r);
    }
  }
  if (!__syncthreads_or(any)) return;
  // workgroup tile: bounding box of its bins' positions +- margin, clipped to [1, N) and to the LDS budget
  if (tid == 0) { s_box[0] = 0x7FFFFFFF; s_box[1] = -0x7FFFFFFF; s_box[2] = 0x7FFFFFFF; s_box[3] = -0x7FFFFFFF; s_reach = 0; s_cnt[0] = 0; s_cnt[1] = 0; }
  __syncthreads();
  {
    // (reduced over the wave first: one atomic per wave, see k_narrow)
    x_lo = wave_mini(x_lo); x_hi = wave_maxi(x_hi); y_lo = wave_mini(y_lo); y_hi = wave_maxi(y_hi);
    const int r_hi = wave_maxi(__float_as_int(reach));       // (non-negative floats order as their bit patterns)
    if ((tid & 63) == 0) {
      atomicMin(&s_box[0], x_lo); atomicMax(&s_box[1], x_hi);
      atomicMin(&s_box[2], y_lo); atomicMax(&s_box[3], y_hi);
      atomicMax(&s_reach, r_hi);
    }
  }
  __syncthreads();
  const float reach_wg = __int_as_float(s_reach);
  bool sure = reach_wg <= (float)kLaneReachMax;               // every electron within `margin` of its bin
  const int margin = sure ? (int)ceilf(reach_wg) : kLaneMargin;
  const int bx0 = s_box[0] - margin, bx1 = s_box[1] + margin + 1, by0 = s_box[2] - margin, by1 = s_box[3] + margin + 1;
  int tx0 = max(bx0, 1), tx1 = min(bx1, a.N), ty0 = max(by0, 1), ty1 = min(by1, a.N);
  int tw = max(tx1 - tx0, 0), th = max(ty1 - ty0, 0);
  sure = sure && s_box[0] <= s_box[1] && tx0 == bx0 && tx1 == bx1 && ty0 == by0 && ty1 == by1 && (long long)tw * th <= kLaneTile;
  while ((long long)tw * th > kLaneTile && th > 1) { ty0 += 1; th = max(th - 2, 1); }
  while ((long long)tw * th > kLaneTile && tw > 1) { tx0 += 1; tw = max(tw - 2, 1); }
  if ((long long)tw * th > kLaneTile) { tw = 0; th = 0; }
  const int tarea = tw * th;
  for (int i = tid; i < tarea; i += kLaneThreads) tile[i] = 0;

  int par = 0;                            // which list counter the current sub-sample uses (THIN)
  // the tile cell at byte offset `addr` gets an electron; THIN: the first one there puts the cell on the flush list
  auto tile_add = [&](int addr) {
    if (THIN) {
      if (atomicAdd((int*)((char*)tile + addr), 1) == 0) {
        const unsigned long long m = __ballot(1);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        int base = 0;
        if (before == 0) base = atomicAdd(&s_cnt[par], __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + before < kLaneListCap) s_list[base + before] = (unsigned short)(addr >> 2);
      }
    } else {
      atomicAdd((int*)((char*)tile + addr), 1);
    }
  };

  for (int k = k0; k < k1; ++k) {
  const size_t kw = (size_t)k * a.W + (inw ? w : 0);
  int n = 0, nw = 0;
  BinLocal bl{};
  if (FUSED) {
    if (inw) {
      const BinPlan b = plan_bin(p, k, w, f_wl, f_flux, f_sens, f_dlam, f_ratio, f_sigl, p.tr + kTrStride * (size_t)k, p.x_ref[k],
                                 p.y_ref[k], p.dur_ms[k], p.depth ? p.depth[kw] : 0.);
      p.counts[kw] = (int32_t)b.count;
      if (b.overflow) atomicOr(p.status, 1);
      if (b.narrow > 0u || b.rest > 0u) atomicOr(p.status, 2);   // a bin for k_narrow / k_throw after all: the host runs the exposure again
      n = (int)b.lane;
      f_electrons += b.lane;
      if (n > 0) { bl = bin_local(b.xs, b.ys); nw = min(max(b.nwide, 0), n); }
    }
  } else if (!BATCH) {
    n = n_one;
  } else if (inw) {
    n = a.nlane[kw];
  }
  if (!__syncthreads_or(n > 0)) continue;                    // (also: the tile is clean and this round's list counter is 0)
  const SubInfo si = FUSED ? s_sub[k - k0] : a.sub[__builtin_amdgcn_readfirstlane(k)];     // (a scalar load: k is the workgroup's)
  if (!FUSED && !BATCH) {
    if (n > 0) { bl = bl_one; nw = nw_one; }
  } else if (!FUSED && n > 0) {
    bl = bin_local(a.xpos[kw], a.ypos[kw]);
    nw = min(max(a.nwide[kw], 0), n);
  }
  // the electron's pixel is (ox, oy) + floor((fx, fy) + its offset): see bin_local.  A lane without electrons in this
  // sub-sample throws at -1e30 (its stream still advances with the wave's)
  const float x = n > 0 ? bl.fx : -1e30f, y = n > 0 ? bl.fy : -1e30f;
  const int ox = bl.ox, oy = bl.oy;

  // Electron j of the bin takes WORD j of the bin's stream (a pair of the stream serves two electrons); the first nw
  // take sigma_h.  One 32-bit word per electron:
  //   angle   = its high half / 2^16 revolutions (+ see `rev`).  65536 equally spaced directions: a pixel's probability is an
  //             integral over the angle of a piecewise smooth periodic function, which an equispaced rule of that
  //             many nodes gives to ~1e-8 -- the position itself moves by < 1e-4 R;
  //   radius  = sigma sqrt(-2 ln u), u = (h + 1/2) / 2^16 from its low half h, the midpoint rule in u -- and where
  //             the cell is not small against the scale on which the radius changes, h = 0 (R > 4.7 sigma, 1.5e-5 of
  //             the electrons), the cell is subdivided by 17 bits h' of a side stream of the bin (a 32-bit LCG seeded from
  //             the bin's Philox block, advanced only here): u = (h' + 1/2) 2^-33, so the radius reaches
  //             sigma sqrt(2 ln 2^34) = 6.87 sigma (it was 6.76 with 32 bits per radius).
  // Half the random words of a pair per electron: 9 of the 23 vector instructions of an electron were the pair.
  // A wave whose bins all had their narrow electrons taken by k_narrow (nw = n everywhere: the usual case) runs with
  // sigma as a loop constant: first the iterations EVERY lane has, with no per-lane test at all (neighbouring bins
  // hold nearly the same number of electrons), then the tail, where the stream still advances in every lane and only
  // the deposit is suppressed for the lanes that are done.  A wave with thin, unsplit bins selects sigma per electron.
  const int tw4 = tw * 4;
  // tile[(oy + j - ty0) * tw + (ox + i - tx0)] as a byte offset from an electron's pixel offsets (i, j) from its bin's pixel
  const int origin = (oy - ty0) * tw4 + (ox - tx0) * 4;
  // (c = -2 ln2 sigma^2, c16 = -16 c: (R sigma)^2 = c log2 u = c (log2(h + 1/2) - 16) as one fma)
  uint32_t refine = 0u;
  auto draw = [&](SeededStream& rng, uint32_t wd, float c, float c16, float px, float py, float& vx, float& vy) {
    // angle in [1, 2) revolutions = the word's high 23 bits as a mantissa (one v_alignbit_b32).  Its 7 lowest bits
    // are the radius half-word's 7 highest: given the radius, the angle still runs over 65536 equally spaced
    // directions -- offset by a fraction of their spacing that depends on the radius
    const float rev = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wd, 9));
    const uint32_t h = wd & 0xFFFFu;
    float r2 = fmaf(c, __builtin_amdgcn_logf((float)h + 0.5f), c16);
    if (__builtin_expect(h == 0u, 0)) {
      refine = refine * 1664525u + 1013904223u;           // (the bin's side stream: leaves the main stream's state alone)
      r2 = fmaf(c, __builtin_amdgcn_logf((float)(refine >> 15) + 0.5f), -33.f * c);
    }
    const float Rs = __builtin_amdgcn_sqrtf(r2);
    vx = fmaf(__builtin_amdgcn_cosf(rev), Rs, px);            // offset + the bin's fraction of a pixel (:91-92; bin_local)
    vy = fmaf(__builtin_amdgcn_sinf(rev), Rs, py);
  };
  // (test-free tile) every live electron is inside the tile and on the frame
  auto throw_sure = [&](SeededStream& rng, uint32_t wd, float c, float c16, bool live) {
    float vx, vy;
    draw(rng, wd, c, c16, x, y, vx, vy);
    // byte address = j * tw4 + (i * 4 + origin), (i, j) = floor(vx, vy) within +-kLaneReachMax: v_cvt_flr_i32_f32 twice,
    // v_lshl_add_u32, v_mad_i32_i24 (the compiler's own choice is a floor, a convert, a multiply, a shift and a
    // three-operand add)
    // (one block: left to itself the compiler takes `origin` apart again and adds its two halves per electron)
    int addr, j;
    asm("v_cvt_flr_i32_f32 %0, %2\n\tv_cvt_flr_i32_f32 %1, %3\n\tv_lshl_add_u32 %0, %0, 2, %4\n\tv_mad_i32_i24 %0, %1, %5, %0"
        : "=&v"(addr), "=&v"(j) : "v"(vx), "v"(vy), "v"(origin), "s"(tw4));
    if (live) tile_add(addr);
  };
  // (tile with the bounds test) the electron's cell counted from the tile's corner: floor of the sum + the bin's pixel
  // relative to that corner, in unsigned arithmetic -- a dead lane's -1e30 saturates the conversion and wraps to a cell
  // off every frame.  As many instructions as the truncating convert and the subtraction of the corner they replace.
  const uint32_t oxl = (uint32_t)ox - (uint32_t)tx0, oyl = (uint32_t)oy - (uint32_t)ty0;
  auto throw_one = [&](SeededStream& rng, uint32_t wd, float c, float c16, float px, float py) {
    float vx, vy;
    draw(rng, wd, c, c16, px, py, vx, vy);
    uint32_t lx, ly;
    asm("v_cvt_flr_i32_f32 %0, %2\n\tv_cvt_flr_i32_f32 %1, %3\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5"
        : "=&v"(lx), "=&v"(ly) : "v"(vx), "v"(vy), "v"(oxl), "v"(oyl));
    if (lx < (uint32_t)tw && ly < (uint32_t)th) {
      tile_add(__umul24(ly, tw4) + (lx << 2));
    } else {
      const int xi = (int)(lx + (uint32_t)tx0), yi = (int)(ly + (uint32_t)ty0);
      if (xi > 0 && xi < a.N && yi > 0 && yi < a.N) deposit_global<FLUSH>(a, si, xi, yi, 1);               // (:93)
    }
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
    refine = (rng.s1 * 0x9E3779B9u) ^ rng.s3;
    const bool one_sigma = !__any(n > nw);
    // electrons [lo, hi) of the lane are thrown; the others of its n take a sigma that is not finite (see bad_h):
    // lo = 0, hi = n in every wave of every exposure this instrument can produce
    const int lo = bad_h ? nw : 0, hi = bad_l ? min(nw, n) : n;
    const bool odd_wave = __any(n > 0 && (bad_h || bad_l));
    const int cmin2 = cmin & ~1;                             // electrons 2i and 2i + 1 share pair i
    const float ch16 = -16.f * ch, cl16 = -16.f * cl;
    uint32_t wa, wb;
    if (odd_wave) {
      // (a wave with such a lane -- its workgroup's tile is never test-free: the reach of a sigma that is not finite is
      // not -- takes ONE general loop; the loops below keep their single test per electron)
      for (int j = 0; j < cmax; j += 2) {
        const bool la = j < hi && j >= lo, lb = j + 1 < hi && j + 1 >= lo;
        rng.next2(wa, wb);
        throw_one(rng, wa, (j < nw) ? ch : cl, (j < nw) ? ch16 : cl16, la ? x : -1e30f, la ? y : -1e30f);
        throw_one(rng, wb, (j + 1 < nw) ? ch : cl, (j + 1 < nw) ? ch16 : cl16, lb ? x : -1e30f, lb ? y : -1e30f);
      }
    } else if (sure) {
      if (one_sigma) {
        for (int j = 0; j < cmin2; j += 2) { rng.next2(wa, wb); throw_sure(rng, wa, ch, ch16, true); throw_sure(rng, wb, ch, ch16, true); }
        for (int j = cmin2; j < cmax; j += 2) { rng.next2(wa, wb); throw_sure(rng, wa, ch, ch16, j < n); throw_sure(rng, wb, ch, ch16, j + 1 < n); }
      } else {
        for (int j = 0; j < cmax; j += 2) {
          rng.next2(wa, wb);
          throw_sure(rng, wa, (j < nw) ? ch : cl, (j < nw) ? ch16 : cl16, j < n);
          throw_sure(rng, wb, (j + 1 < nw) ? ch : cl, (j + 1 < nw) ? ch16 : cl16, j + 1 < n);
        }
      }
    } else if (one_sigma) {
      for (int j = 0; j < cmin2; j += 2) { rng.next2(wa, wb); throw_one(rng, wa, ch, ch16, x, y); throw_one(rng, wb, ch, ch16, x, y); }
      for (int j = cmin2; j < cmax; j += 2) {
        const bool la = j < n, lb = j + 1 < n;
        rng.next2(wa, wb);
        throw_one(rng, wa, ch, ch16, la ? x : -1e30f, la ? y : -1e30f);
        throw_one(rng, wb, ch, ch16, lb ? x : -1e30f, lb ? y : -1e30f);
      }
    } else {
      for (int j = 0; j < cmax; j += 2) {
        const bool la = j < n, lb = j + 1 < n;
        rng.next2(wa, wb);
        throw_one(rng, wa, (j < nw) ? ch : cl, (j < nw) ? ch16 : cl16, la ? x : -1e30f, la ? y : -1e30f);
        throw_one(rng, wb, (j + 1 < nw) ? ch : cl, (j + 1 < nw) ? ch16 : cl16, lb ? x : -1e30f, lb ? y : -1e30f);
      }
    }
  }
  __syncthreads();
  // flush with THIS sub-sample's flat, leaving the tile clean
  const int listed = THIN ? s_cnt[par] : 0;
  if (THIN && listed <= kLaneListCap) {
    for (int t = tid; t < listed; t += kLaneThreads) {
      const int i = (int)s_list[t];
      const int m = tile[i];
      tile[i] = 0;
      const int ly = i / tw, lx = i - ly * tw;
      deposit_global<FLUSH>(a, si, tx0 + lx, ty0 + ly, m);
    }
  } else {
    for (int i = tid; i < tarea; i += kLaneThreads) {
      const int m = tile[i];
      if (m > 0) {
        tile[i] = 0;
        const int ly = i / tw, lx = i - ly * tw;
        deposit_global<FLUSH>(a, si, tx0 + lx, ty0 + ly, m);
      }
    }
  }
  // The counter of the NEXT round is cleared here, not this round's: every thread reads this round's count after the
  // barrier above and nothing orders a late reader before a reset by thread 0 -- a wave that read 0 would skip its
  // share of the list and leave cells that never return 0 on first touch again.  The other counter was last read in
  // the previous round's flush, which every thread left before this round's first barrier.
  if (THIN && tid == 0) s_cnt[par ^ 1] = 0;
  par ^= 1;
  }   // sub-samples of the batch
  if (FUSED) {
    // electrons handed to the lanes: ONE atomic per workgroup, on the workgroup's stripe of the counter (count_electrons;
    // one per wave on a single address was 1.4 ms of a finely sampled exposure)
    for (int off = 32; off > 0; off >>= 1) f_electrons += __shfl_down(f_electrons, off);
    __shared__ unsigned long long s_tot[kLaneThreads / 64];
    if ((tid & 63) == 0) s_tot[tid >> 6] = f_electrons;
    __syncthreads();
    if (tid == 0) {
      unsigned long long t = 0;
      for (int i = 0; i < kLaneThreads / 64; ++i) t += s_tot[i];
      if (t) count_electrons(p.total_electrons, blockIdx.y * gridDim.x + blockIdx.x, t);
    }
  }
}

template <int FLUSH, bool THIN, bool BATCH>
__global__ __launch_bounds__(kLaneThreads) void k_lane(ThrowArgs a) {
  lane_body<FLUSH, THIN, BATCH, false>(a, PrepArgs{}, CosmicArgs{});
}
// (two workgroups per CU: 128 VGPRs)
template <int FLUSH>
__global__ __launch_bounds__(kLaneThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lane_fused(ThrowArgs a, PrepArgs p, CosmicArgs ca) {
  lane_body<FLUSH, true, true, true>(a, p, ca);
}

}  // namespace wayne
