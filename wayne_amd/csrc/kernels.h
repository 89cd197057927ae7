// HIP kernels (gfx950) of the WFC3-IR exposure-synthesis path, one header per stage:
//
//   k_lightcurve.h  k_lightcurve   transit depths depth[K][W] from z[K], rp[W], limb darkening
//   k_prep.h        k_prep_wl      per-wavelength arrays: PSF polynomials, sensitivity LUT, bin widths   (A8, A9)
//                   k_prep_sub     per sub-sample: trace, bin positions, expected counts, Poisson/round,
//                                  sigma split, routing of the bins, prefix, electron count and clip rectangle
//                                  per sub-sample; cosmic-ray hits per read interval          (A6, A7, A9, A10, A13, cosmic_rays.py)
//   k_throw.h       k_throw        the electron thrower: LDS int32 tile per workgroup slice,
//                                  flushed x flat into the read-interval accumulator                     (A1-A4, A11, A12)
//   k_narrow.h      k_narrow       narrow PSF component as one multinomial per bin
//                   k_lane         a bin's one-by-one electrons (wide component, thin bins) thrown by its own lane
//                   k_lane_fused   thin exposures: k_lane planning its bins itself (plan_bin of k_prep.h), no k_prep_sub
//   k_ramp.h        k_ramp         fused up-the-ramp kernel: sky, gain, cumulative, dark, non-linearity,
//                                  clip, reference pixels, zero read, read noise                          (A13-A15)
//
// "A<n>" are the row ids of SURVEY.md section 8(a); reference file:line
// citations are next to each formula.
#pragma once
#include "common.h"
#include "k_lightcurve.h"
#include "k_prep.h"
#include "k_throw.h"
#include "k_narrow.h"
#include "k_ramp.h"
