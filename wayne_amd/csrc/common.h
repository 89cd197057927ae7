// Shared constants and structures of the gfx950 kernels (included through kernels.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "philox.h"
#include "plan_consts.h"
#include "samplers.h"

namespace wayne {

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
// (GrismDev, kBorder and the routing / launch-shape constants the host planner shares: plan_consts.h)

// Per sub-sample record written by k_prep_sub and read by k_throw.
constexpr int kTrStride = 8;   // doubles per sub-sample in the trace-coefficient array (k_prep_wl): 6 coefficients, 1 / m_wl, pad

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

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------
// bin-local coordinates of the production throwers (k_lane, k_throw's Philox mode, k_narrow)
// ---------------------------------------------------------------------------
// The reference adds an electron's offset to its bin's position in fp64 and truncates (pyparallel_menu.c:91-92).  The
// production throwers draw the offset in float32; adding it to the bin's FRAME coordinate in float32 would round the
// sum on the frame's scale (half an ulp at x >= 512: 3e-5 px, twice: in the cast of the position and in the add).
// Instead the position is split in fp64, once per bin, into its pixel and the fraction inside it,
//     pos = o + f,  o = floor(pos) (an integer),  f = (float)(pos - o) in [0, 1],
// the float32 sum is f + offset -- rounded on the scale of the electron's DISTANCE from its bin (1e-7 px within a
// pixel, 5e-7 px at 6 px, 4e-6 px at the 48 px a wide electron can reach) -- and the electron's pixel is
//     o + floor(f + offset).
// floor(), not the reference's truncation toward zero: the two differ only for sums in (-1, 0) of the frame coordinate,
// which truncation sends to pixel 0 and floor to pixel -1 -- both outside the reference's strict bounds 0 < pos < n
// (:93), so the kept electrons are the same.  A position that is not finite or beyond +-1e6 is not split (o = 0,
// f = (float)pos): with any PSF of the instrument its electrons miss the frame either way.  A sigma that is not finite
// (or whose square overflows a float) is settled where the bin is loaded: its electrons are thrown at -1e30, off every
// frame like the reference's (int) of a non-finite double -- so the sums in the electron loops are finite.
// oracle/split_oracle.c (so_bin_local) and oracle/psf_oracle.c are the same statement on the CPU.
struct BinLocal { float fx, fy; int ox, oy; bool sane; };
__device__ __forceinline__ BinLocal bin_local(double xd, double yd) {
  BinLocal b;
  const bool sane = fabs(xd) < 1e6 && fabs(yd) < 1e6;
  b.sane = sane;
  const double flx = sane ? floor(xd) : 0., fly = sane ? floor(yd) : 0.;
  b.fx = (float)(xd - flx); b.fy = (float)(yd - fly);
  b.ox = (int)flx; b.oy = (int)fly;
#ifdef WAYNE_NEGCTL_DROP_FRACTION
  // NEGATIVE-CONTROL BUILD (wrong frames; tests/test_visit_science_gpu.py): the production throwers forget where inside
  // its pixel a bin sits -- the gross form of a position-rounding defect, to show that the visit-level measurement sees one
  b.fx = 0.f; b.fy = 0.f;
#endif
  return b;
}
// (int)floorf(v) as ONE instruction (v_cvt_flr_i32_f32: floor, convert, saturate; NaN gives 0) -- the compiler's own
// choice for the expression is v_floor_f32 + v_cvt_i32_f32
__device__ __forceinline__ int floor_to_int(float v) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}


// Reductions over the 16 lanes of a DPP row (butterfly of row rotations): every lane of the row gets the result.
// All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ int dpp_row(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
#define WAYNE_ROW16(name, T, cast_in, cast_out, op)                                   \
  __device__ __forceinline__ T name(T v) {                                            \
    v = op(v, cast_out(dpp_row<0x128>(cast_in(v))));  /* row_ror:8 */                 \
    v = op(v, cast_out(dpp_row<0x124>(cast_in(v))));  /* row_ror:4 */                 \
    v = op(v, cast_out(dpp_row<0x122>(cast_in(v))));  /* row_ror:2 */                 \
    v = op(v, cast_out(dpp_row<0x121>(cast_in(v))));  /* row_ror:1 */                 \
    return v;                                                                         \
  }
__device__ __forceinline__ int wayne_addi(int a, int b) { return a + b; }
WAYNE_ROW16(row16_min, float, __float_as_int, __int_as_float, fminf)
WAYNE_ROW16(row16_max, float, __float_as_int, __int_as_float, fmaxf)
WAYNE_ROW16(row16_mini, int, , , min)
WAYNE_ROW16(row16_maxi, int, , , max)
WAYNE_ROW16(row16_sum, int, , , wayne_addi)
#undef WAYNE_ROW16
// ... and over the whole wave (wave-uniform result)
__device__ __forceinline__ int wave_mini(int v) {
  v = row16_mini(v);
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_maxi(int v) {
  v = row16_maxi(v);
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// the sum of one 32-bit number per lane, as 64 bits (two 16-lane sums of 24 + 8 bits each: no carry to lose)
__device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t n) {
  const int lo = row16_sum((int)(n & 0xFFFFFFu)), hi = row16_sum((int)(n >> 24));
  const unsigned long long L = (unsigned long long)(uint32_t)(__builtin_amdgcn_readlane(lo, 0) + __builtin_amdgcn_readlane(lo, 16) +
                                                              __builtin_amdgcn_readlane(lo, 32) + __builtin_amdgcn_readlane(lo, 48));
  const unsigned long long H = (unsigned long long)(uint32_t)(__builtin_amdgcn_readlane(hi, 0) + __builtin_amdgcn_readlane(hi, 16) +
                                                              __builtin_amdgcn_readlane(hi, 32) + __builtin_amdgcn_readlane(hi, 48));
  return L + (H << 24);
}
#endif

}  // namespace wayne
