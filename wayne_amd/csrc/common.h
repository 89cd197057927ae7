// Shared constants and structures of the gfx950 kernels (included through kernels.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "philox.h"
#include "samplers.h"

namespace wayne {

constexpr int kBorder = 5;            // reference-pixel border (detector.py:146-147)
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
struct GrismDev {
  double trace[9], wlsol[9];
  double p_ratio[4], p_sigl[4], p_sigh[4];
  double flat_wmin, flat_wmax;
  int n_sens;
  const double* sens_wl;
  const double* sens_val;
};

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

}  // namespace wayne
