// Constants and small structures shared by the gfx950 kernels AND the host-side launch planner (host_plan.h).
// No HIP header is included here: this file and host_plan.h also compile with plain g++ (tests/native/plan_harness.cpp
// builds the planner under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU).
#pragma once
#include <math.h>
#include <stdint.h>
#include "philox.h"   // WAYNE_HD

namespace wayne {

constexpr int kBorder = 5;            // reference-pixel border (detector.py:146-147)

// What grism.G141 / grism.G102 hold (grism.py:24-118, 426-476, 756-776), as the kernels take it.
struct GrismDev {
  double trace[9], wlsol[9];
  double p_ratio[4], p_sigl[4], p_sigh[4];
  double flat_wmin, flat_wmax;
  int n_sens;
  const double* sens_wl;
  const double* sens_val;
};

// ---- routing of a bin's electrons (k_prep_sub decides, the host estimates: both from these numbers) ----
constexpr int kSplitMin = 32;          // WAYNE_RNG_SPLIT: bins with fewer narrow electrons are thrown one by one
constexpr int kNarrowR = 6;            // k_narrow window: +-6 pixels about the bin's pixel (>= 6.5 sigma_l)
constexpr int kLaneMax = 4096;         // WAYNE_RNG_SPLIT: a bin's one-by-one electrons are thrown by its own lane (k_lane) up to this many
constexpr int kLaneReach = 64 * kLaneMax;   // ... and up to this many in an exposure launched without k_throw (a lane then needs
                                       // ~10 ms for its bin; beyond it the exposure is run again with k_throw)
constexpr uint32_t kSplitMaxNarrow = 1u << 24;   // k_narrow's chain counts in float32: larger bins are thrown one by one

// ---- launch shapes ----
constexpr int kNarrowThreads = 512;     // bins per k_narrow workgroup
constexpr int kLaneThreads = 512;       // bins per k_lane workgroup
constexpr int kMaxChunks = 128;         // >= 32768 bins / bins per k_narrow / k_lane workgroup
constexpr int kLaneListCap = 4096;      // cells on a THIN flush list (beyond it the flush falls back to the scan)
constexpr int kLaneBatchMax = 32;       // most sub-samples per k_lane workgroup

// ---- sky draw of k_ramp ----
constexpr int kSkyAlias = 256;    // entries per alias table: alias << 24 | 24-bit acceptance threshold
constexpr float kSkyPiece = 16.f; // largest mean drawn by one sequential search
constexpr int kMaxReads = 15;     // NSAMP <= 16 (detector.py:228)

WAYNE_HD void trace_coeffs(const GrismDev& g, double x_ref, double y_ref, double* o) {
  // o = {m_t, c_t, m_w, c_w, m_wl, c_wl}
    // wavelength_calibration_coeffs (grism.py:779-803)
    const double* t = g.trace;
    const double* b = g.wlsol;
    const double m_t = t[3] + t[4] * x_ref + t[5] * y_ref + t[6] * (x_ref * x_ref) +
                       t[7] * x_ref * y_ref + t[8] * (y_ref * y_ref);
    const double c_t = t[0] + t[1] * x_ref + t[2] * y_ref;
    const double m_w = b[3] + b[4] * x_ref + b[5] * y_ref + b[6] * (x_ref * x_ref) +
                       b[7] * x_ref * y_ref + b[8] * (y_ref * y_ref);
    const double c_w = (b[0] + b[1] * x_ref) + b[2] * y_ref;
    // _get_x_to_wl_poly_coeffs (grism.py:553-602): line through the trace
    // points at x_ref+10 and x_ref+20, wavelength in micron.
    const double xa = x_ref + 10, xb = x_ref + 20;
    const double ya = m_t * (xa - x_ref) + c_t + y_ref;  // x_to_y (grism.py:537)
    const double yb = m_t * (xb - x_ref) + c_t + y_ref;
    const double da = sqrt((ya - y_ref) * (ya - y_ref) + (xa - x_ref) * (xa - x_ref));
    const double db = sqrt((yb - y_ref) * (yb - y_ref) + (xb - x_ref) * (xb - x_ref));
    const double wa_ = (m_w * da + c_w) * 1e-4;  // angstrom -> micron
    const double wb_ = (m_w * db + c_w) * 1e-4;
    const double m_wl = (wb_ - wa_) / (xb - xa);
    const double c_wl = wa_ - m_wl * xa;
    o[0] = m_t; o[1] = c_t; o[2] = m_w; o[3] = c_w; o[4] = m_wl; o[5] = c_wl;
}

}  // namespace wayne
