#!/usr/bin/env python3
"""Mutation audit of the GPU suite: plant ONE defect at a time in a copy of the library's sources, build it, and ask
which test notices.

    python scripts/mutation_audit.py build            # here (hipcc cross-compiles): tests/native/_build/mutants/<name>.so
    python scripts/mutation_audit.py run [name ...]   # on the GPU box: gpurun_out/mutation_audit.txt
    python scripts/mutation_audit.py cpu [name ...]   # here: host-only mutants (FITS writer) against the CPU suite

Why.  The oracle restates the reference's algorithm, and for three rounds it shared a defect with the kernels it checks
(the sky draw's runaway search): same-counter parity was green with ~500 spurious electrons in one pixel per exposure.
The question a reviewer cannot answer from a green suite is "what would you NOT see?".  So every mutant is run against
two sets of tests, in this order:

  INDEPENDENT  tests that do not go through the oracle's restatement of the mutated stage: the exact laws read off the
               reference's text (test_extremes_gpu, test_detector_laws_gpu), stream independence (test_independence_gpu),
               ensembles of the COMPILED reference C (test_ensemble_gpu), the reference's own golden frames
               (test_psf_gpu's golden test), size-independent properties (test_configs_gpu).  A mutant killed here would have been
               caught even if the oracle had shared the defect.
  REST         the rest of the -m gpu suite (same-counter parity against the oracle): run only for a mutant the first set
               let through -- such a mutant names a stage whose ONLY guard is the restatement.

A mutant is a list of (file under wayne_amd/csrc, exact text, replacement); the text must occur exactly once.  The
shipped sources are never touched: each mutant is built from a copy under tests/native/_build/mutants/ and loaded
through WAYNE_HIP_LIB (wayne_amd/_lib.py), like the negative-control libraries.  Test infrastructure; results of round 5:
profiles/r05/mutation_audit.txt, HISTORY.md section 6.
"""
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import build as wb  # noqa: E402

OUT_DIR = os.path.join(ROOT, "tests", "native", "_build", "mutants")
REPORT = os.path.join(ROOT, "gpurun_out", "mutation_audit.txt")

INDEPENDENT = ["tests/test_detector_laws_gpu.py", "tests/test_independence_gpu.py", "tests/test_extremes_gpu.py",
               "tests/test_psf_gpu.py::test_replay_mode_is_bit_exact_against_reference_golden",
               "tests/test_ensemble_gpu.py", "tests/test_configs_gpu.py"]

MUTANTS = [
    # --- detector stages of k_ramp (A13-A15)
    dict(name="read_noise_5pc", stage="A15 read noise (detector.py:33, 193-198)",
         what="read noise 14.8 / 2.35 instead of 14.1 / 2.35",
         edits=[("common.h", "constexpr double kReadNoise = 14.1 / 2.35;", "constexpr double kReadNoise = 14.8 / 2.35;")]),
    dict(name="gain_2pc", stage="A13 gain (detector.py:30, 203-204; exposure_generator.py:507-511)",
         what="gain 2.40 instead of 2.35 (constant and per-pixel form)",
         edits=[("common.h", "constexpr double kGain = 2.35;", "constexpr double kGain = 2.40;"),
                ("k_ramp.h", "const float g32 = 2.35f / t_pfl;", "const float g32 = 2.40f / t_pfl;")]),
    dict(name="clip_max", stage="A15 clip (detector.py:26-28)",
         what="upper detector limit 77 000 instead of 78 000 DN",
         edits=[("common.h", "constexpr double kMaxCounts = 78000.0;", "constexpr double kMaxCounts = 77000.0;")]),
    dict(name="dark_err_floor", stage="A15 dark (detector.py:189-190: err <= 0 -> 1e-5)",
         what="non-positive dark errors replaced by 0.01 instead of 1e-5",
         edits=[("wayne_hip.hip", "if (!(x > 0.f)) x = 0.00001f;", "if (!(x > 0.f)) x = 0.01f;"),
                ("k_ramp.h", "const double err = (de > 0.f) ? (double)de : (double)0.00001f;",
                 "const double err = (de > 0.f) ? (double)de : (double)0.01f;")]),
    dict(name="nonlinear_c3_c4", stage="A15 non-linearity (detector.py:335-348)",
         what="cubic and quartic coefficient planes exchanged in the production (float32) solve",
         edits=[("k_ramp.h", "const float h = fmaf(u, fmaf(u, fmaf(u, c4, c3), c2), c1p);",
                 "const float h = fmaf(u, fmaf(u, fmaf(u, c3, c4), c2), c1p);")]),
    dict(name="reference_pixels_kept", stage="A15 reference pixels (exposure.py:122-131)",
         what="reference pixels not reset to zero in the production chain",
         edits=[("k_ramp.h", "if (!interior) v = 0.f;                            // reference pixels (exposure.py:122-131)",
                 "if (false) v = 0.f;")]),
    dict(name="zero_read_noiseless", stage="A14 zero read (exposure.py:61-68)",
         what="no read noise on the zero read",
         edits=[("k_ramp.h", "if (rdn) { bm_pair<FAST>(w0, w1, zd, zr); v = v + kReadNoise * (double)zr; }",
                 "if (rdn) { bm_pair<FAST>(w0, w1, zd, zr); v = v + 0.0 * (double)zr; }")]),
    # --- sky (A13)
    dict(name="sky_remainder_pmf", stage="A13 sky Poisson (exposure_generator.py:488-495)",
         what="third term of the remainder's pmf m^2 * 0.6 instead of m^2 / 2 (production integer thresholds)",
         edits=[("k_ramp.h", "t1 = thr(cdf);  t = t * (m_ * 0.5f);         cdf += t;",
                 "t1 = thr(cdf);  t = t * (m_ * 0.6f);         cdf += t;")]),
    dict(name="sky_shared_stream", stage="A13 / A15 random streams (exposure index in the key)",
         what="the per-pixel read stream of k_ramp keyed without the exposure index: every exposure draws the same noise",
         edits=[("k_ramp.h", "rn = SeededStream(a.seed, STAGE_READ, (uint32_t)p, 0u, a.exposure);",
                 "rn = SeededStream(a.seed, STAGE_READ, (uint32_t)p, 0u, 0u);")]),
    # --- cosmic rays (A13)
    dict(name="cosmic_energy_range", stage="A13 cosmic rays (cosmic_rays.py:127-134)",
         what="hit energies randint(20000, 45000) instead of (10000, 35000)",
         edits=[("k_prep.h", "const uint32_t energy = 10000u + uint_below(w.v[0], 25000u);",
                 "const uint32_t energy = 20000u + uint_below(w.v[0], 25000u);")]),
    # --- counts chain / flat / trace (A9-A11)
    dict(name="counts_chain_1pc", stage="A9 counts chain (exposure_generator.py:602-628)",
         what="the 1e4 A / um factor of the counts chain 1 % high",
         edits=[("k_prep.h", "  lam = lam * 1e4;", "  lam = lam * 1.01e4;")]),
    dict(name="flat_cubic_terms", stage="A11 flat (grism.py:362-385)",
         what="quadratic and cubic flat planes exchanged",
         edits=[("k_throw.h", "((double)a.flat[2][i] * t2) +\n                   ((double)a.flat[3][i] * t3);",
                 "((double)a.flat[2][i] * t3) +\n                   ((double)a.flat[3][i] * t2);")]),
    # --- thrower (A1-A4)
    dict(name="wide_sigma_2pc", stage="A4 wide PSF component (pyparallel_menu.c:87-108)",
         what="sigma_h of the lane-thrown wide electrons 2.3 % large (k_lane)",
         edits=[("k_narrow.h", "    ch = (-1.3862943611198906f * sh) * sh;", "    ch = (-1.45f * sh) * sh;")]),
    dict(name="narrow_cell_masses", stage="A4 narrow PSF component as multinomials",
         what="cell masses of the multinomial from a gaussian 1.8 % narrow (argument scale of the tail fit)",
         edits=[("k_narrow.h", "const float z = t * 0.70710678118654752f;", "const float z = t * 0.72f;")]),
    dict(name="lane_cos_sin_swapped", stage="A2 / A4 Box-Muller (pyparallel_menu.c:91-93)",
         what="k_lane takes x from the sine and y from the cosine: the same law, other electrons",
         edits=[("k_narrow.h", "    vx = fmaf(__builtin_amdgcn_cosf(rev), Rs, px);            // offset + the bin's fraction of a pixel (:91-92; bin_local)\n"
                               "    vy = fmaf(__builtin_amdgcn_sinf(rev), Rs, py);",
                 "    vx = fmaf(__builtin_amdgcn_sinf(rev), Rs, px);\n"
                 "    vy = fmaf(__builtin_amdgcn_cosf(rev), Rs, py);")]),
    dict(name="bin_fraction_dropped", stage="A4 position of a bin inside its pixel (pyparallel_menu.c:91-92; common.h bin_local)",
         what="the production throwers forget the fraction of a pixel of every bin's position: electrons thrown from the pixel's corner",
         edits=[("common.h", "  b.fx = (float)(xd - flx); b.fy = (float)(yd - fly);", "  b.fx = 0.f; b.fy = 0.f;")]),
    dict(name="lane_floor_is_trunc", stage="A4 the electron's pixel (pyparallel_menu.c:91-93; common.h bin_local: floor of fraction + offset)",
         what="k_lane's test-free path truncates the local sum toward zero: electrons left of / below their bin's pixel land one pixel high",
         edits=[("k_narrow.h", 'asm("v_cvt_flr_i32_f32 %0, %2\\n\\tv_cvt_flr_i32_f32 %1, %3\\n\\tv_lshl_add_u32',
                 'asm("v_cvt_i32_f32 %0, %2\\n\\tv_cvt_i32_f32 %1, %3\\n\\tv_lshl_add_u32')]),
    dict(name="settle_never_reruns", stage="host: wayne_ctx_synchronize settles incomplete exposures (include/wayne_hip.h)",
         what="wayne_ctx_synchronize reads the status words and never runs a flagged exposure again",
         edits=[("wayne_hip.hip", "} else if ((status & 2) && pass == 0) {", "} else if (false && (status & 2) && pass == 0) {")]),
    dict(name="wide_count_rounded", stage="A4 N = (int)(counts ratio) (pyparallel_menu.c:89)",
         what="the wide count rounded to nearest instead of truncated (thrower call and exposure path)",
         edits=[("host_plan.h", "                                                   : (int32_t)nw;",
                 "                                                   : (int32_t)(nw + 0.5);"),
                ("k_prep.h", ": (int32_t)nw;", ": (int32_t)(nw + 0.5);")]),
    # --- host planner
    dict(name="boxes_too_small", stage="host planner: accumulator boxes of k_ramp",
         what="reach of the accumulator boxes 3 sigma instead of 6.9 sigma: k_ramp never loads the electrons beyond",
         edits=[("host_plan.h", "const double reach = 6.9 * e.smax + 2. + 1.;", "const double reach = 3.0 * e.smax + 2. + 1.;")]),
    dict(name="read_trigger", stage="A12 read trigger (exposure_generator.py:336-378)",
         what="every sub-sample accumulates into the read interval before its own (the first into its own)",
         edits=[("k_prep.h", "  si.read = a.sample_read[k];", "  si.read = max(a.sample_read[k] - 1, 0);")]),
    # --- second wave
    dict(name="dark_read_shifted", stage="A15 dark (detector.py:183-190: the frame of read NSAMP index n)",
         what="read 1 takes the dark frame of read 2 (its error plane stays its own)",
         edits=[("k_ramp.h", "float ds_next = ld_dark ? ld_f32(rs_ds, 0) : 0.f, de_next = ld_dark ? ld_f32(rs_de, 0) : 0.f;",
                 "float ds_next = ld_dark ? ld_f32(rs_ds, 1) : 0.f, de_next = ld_dark ? ld_f32(rs_de, 0) : 0.f;")]),
    dict(name="nonlinear_f64_c3_c4", stage="A15 non-linearity, float64 chain",
         what="cubic and quartic coefficients exchanged in the float64 solve's residual",
         edits=[("k_ramp.h", "const double f = fma(u0, fma(u0, fma(u0, fma(k4, u0, k3), k2), k1), -px);",
                 "const double f = fma(u0, fma(u0, fma(u0, fma(k3, u0, k4), k2), k1), -px);")]),
    dict(name="sens_nearest_lower", stage="A8 sensitivity (grism.py:116-118: np.interp)",
         what="the sensitivity table read at the entry below the wavelength instead of interpolated",
         edits=[("k_prep.h", "    s = slope * (x - g.sens_wl[lo]) + g.sens_val[lo];", "    s = 0. * slope + g.sens_val[lo];")]),
    dict(name="dlam_end_bins", stage="A9 bin widths (tools.py:106-128)",
         what="the end bins take one half-gap instead of mirroring it",
         edits=[("k_prep.h", "    left = (i == 0) ? (wl[1] - wl[0]) / 2. : (wl[i] - wl[i - 1]) / 2.;",
                 "    left = (i == 0) ? 0. : (wl[i] - wl[i - 1]) / 2.;")]),
    dict(name="sigl_from_sigh_poly", stage="A8 PSF polynomials (grism.py:85-90)",
         what="sigma_l evaluated with the polynomial of sigma_h",
         edits=[("k_prep.h", "  o.sigl[i] = poly3(g.p_sigl, x);", "  o.sigl[i] = poly3(g.p_sigh, x);")]),
    dict(name="trace_cross_term", stage="A6 / A7 trace (grism.py:779-803)",
         what="the x y term of the trace's slope polynomial dropped",
         edits=[("plan_consts.h", "                       t[7] * x_ref * y_ref + t[8] * (y_ref * y_ref);\n    const double c_t",
                 "                       0. * t[7] * x_ref * y_ref + t[8] * (y_ref * y_ref);\n    const double c_t")]),
    dict(name="wl_solution_cross_term", stage="A6 / A7 dispersion solution (grism.py:779-803)",
         what="the x y term of the dispersion polynomial dropped",
         edits=[("plan_consts.h", "                       b[7] * x_ref * y_ref + b[8] * (y_ref * y_ref);",
                 "                       0. * b[7] * x_ref * y_ref + b[8] * (y_ref * y_ref);")]),
    dict(name="alias_table_bias", stage="A13 sky: Walker tables of the host planner",
         what="the alias construction takes 0.995 instead of 1 off a donor column",
         edits=[("host_plan.h", "    q[l_] = (q[l_] + q[s_]) - 1.;", "    q[l_] = (q[l_] + q[s_]) - 0.995;")]),
    dict(name="cosmic_rate_unscaled", stage="A13 cosmic rays (cosmic_rays.py:33-44: rate per 1024^2 scaled to the frame)",
         what="the hit rate not scaled to the frame's size",
         edits=[("k_prep.h", "const double rate_size = a.rate / (1024. * 1024.) * (double)((long long)a.N * a.N);",
                 "const double rate_size = a.rate;")]),
    dict(name="lc_limb_exponent", stage="f3 light curves (observation.py:293-357; Claret law)",
         what="the third limb-darkening term with mu^2 instead of mu^(3/2) in the quadrature",
         tests=["tests/test_lightcurve.py"],
         edits=[("k_lightcurve.h", "- a3 * (1.f - mu * sm) - a4", "- a3 * (1.f - mu * mu) - a4")]),
    dict(name="lc_eclipse_norm", stage="f3 eclipse term (observation.py:352-355)",
         what="the eclipse term without its 1 / (1 + f) normalisation",
         tests=["tests/test_lightcurve.py"],
         edits=[("k_lightcurve.h", "      ecl = f * hid / (1. + f);", "      ecl = f * hid;")]),
    dict(name="lane_no_tail_refinement", stage="A2 / A4 wide electrons beyond 4.85 sigma (k_lane's one-word draw)",
         what="the radius cell h = 0 not subdivided: 1.5e-5 of the wide electrons land AT 4.855 sigma, none beyond",
         edits=[("k_narrow.h", "if (__builtin_expect(h == 0u, 0)) {", "if (false) {")]),
    dict(name="narrow_tail_cut", stage="A4 narrow component: where the multinomial's window ends",
         what="the gaussian tail taken as 0 beyond 4 sigma_l instead of 6.5",
         edits=[("k_narrow.h", "constexpr float kTailCut = 6.5f;", "constexpr float kTailCut = 4.0f;")]),
    dict(name="sky_remainder_capped", stage="A13 sky Poisson: the remainder beyond three",
         what="the remainder's search stops at 3 (no fourth compare, no continuation)",
         edits=[("k_ramp.h", "  if (wr > sr.t3) {\n    k += 1;", "  if (false) {\n    k += 1;")]),
    dict(name="ptrs_quick_accept", stage="A13 sky (direct sampler) and cosmic-ray counts: PTRS, Hoermann's transformed rejection (samplers.h)",
         what="the quick-acceptance region of a trial widened (us >= 0.03 instead of 0.07): no density test where one is due",
         edits=[("samplers.h", "    if (us >= (T)0.07 && V <= vr) return 1;", "    if (us >= (T)0.03 && V <= vr) return 1;")]),
    dict(name="stellar_ptrs_quick_accept", stage="A9 stellar Poisson noise: k_prep_sub's own PTRS (fp64 behind an fp32 squeeze)",
         what="the quick-acceptance region of a trial widened (us >= 0.03 instead of 0.07) in the stellar counts' sampler",
         edits=[("k_prep.h", "    if (us >= 0.07 && V <= vr) return k;", "    if (us >= 0.03 && V <= vr) return k;")]),
    dict(name="btrs_quick_accept", stage="A4 narrow component: BTRS (Hoermann's binomial transformed rejection) in k_narrow's chains",
         what="the quick-acceptance region of a BTRS trial widened (us >= 0.03 instead of 0.07)",
         edits=[("samplers.h", "      if (us >= (T)0.07 && V <= vr) { x = k; break; }", "      if (us >= (T)0.03 && V <= vr) { x = k; break; }")]),
    # --- third wave
    dict(name="throw_sigma_swapped", stage="A4 per-electron thrower: which electrons take the wide gaussian (pyparallel_menu.c:89-98)",
         what="k_throw gives the first n_wide electrons of a bin the NARROW sigma and the rest the wide one",
         edits=[("k_throw.h", "              c = wide ? cur.ch : cur.cl;", "              c = wide ? cur.cl : cur.ch;")]),
    dict(name="replay_lcg_increment", stage="A2 rand_r replay (glibc: next = next * 1103515245 + 12345)",
         what="the LCG's increment 12346 in the replay thrower's sequential step",
         edits=[("k_throw.h", "  s = s * 1103515245u + 12345u; r = (s >> 16) & 2047u;", "  s = s * 1103515245u + 12346u; r = (s >> 16) & 2047u;")]),
    dict(name="pooled_common_mass", stage="A4 narrow component: pooled rows, the mass Z the group shares",
         what="the common mass Z of a pooling group 2 % low (more electrons go the residual way, with the residual law of the true Z)",
         edits=[("k_narrow.h", "      Z = fminf(PL[kNarrowR] + SU[kNarrowR + 1], 1.f);", "      Z = 0.98f * fminf(PL[kNarrowR] + SU[kNarrowR + 1], 1.f);")]),
    dict(name="binv_cap_8", stage="A4 narrow component: BINV (inversion for n p < 10)",
         what="the inversion search gives up after 8 steps and returns the mean",
         edits=[("samplers.h", "    for (int it = 0; it < 64; ++it) {\n      if (u <= r) break;", "    for (int it = 0; it < 8; ++it) {\n      if (u <= r) break;"),
                ("samplers.h", "    if (x >= (T)64 && x < n) x = M::floor_(n * p + (T)0.5);", "    if (x >= (T)8 && x < n) x = M::floor_(n * p + (T)0.5);")]),
]


# Mutants of the HOST half (Python: sample loop, read trigger, frame offset, scan positions -- SURVEY 8 row A12): edits are
# relative to the repository root and the tests run in a temporary copy of the tree (the shipped library as it is).
PY_MUTANTS = [
    dict(name="py_scan_speed", stage="A12 scan positions (exposure_generator.py:247, 258)",
         what="scan speed 1 % high in the sub-samples' y positions",
         edits=[("wayne_amd/exposure_generator.py", "scan_speed_ms = scan_speed / 1000.          # px/s -> px/ms (:247)",
                 "scan_speed_ms = scan_speed / 1000. * 1.01")]),
    dict(name="py_read_trigger", stage="A12 read trigger (`if i in read_index`, exposure_generator.py:361)",
         what="the sub-sample that closes a read is counted into the next read",
         edits=[("wayne_amd/exposure_generator.py", 'np.searchsorted(np.asarray(read_index), np.arange(K), side="left")',
                 'np.minimum(np.searchsorted(np.asarray(read_index), np.arange(K), side="right"), R - 1)')]),
    dict(name="py_sub_scale", stage="A10 sub-array offset (exposure_generator.py:630)",
         what="frame offset 512 - SUBARRAY / 2 instead of 507 - SUBARRAY / 2",
         edits=[("wayne_amd/exposure_generator.py", "sub_scale = 507 - self.SUBARRAY // 2", "sub_scale = 512 - self.SUBARRAY // 2")]),
    dict(name="py_read_dt", stage="A13 read intervals (exposure_generator.py:362-365)",
         what="every read interval as long as the first",
         edits=[("wayne_amd/exposure_generator.py", "read_dt = np.diff(np.concatenate([[0.0], self.read_times]))                   # (:362-365)",
                 "read_dt = np.full(len(self.read_times), float(self.read_times[0]))")]),
    dict(name="py_ssv_amplitude", stage="A12 scan speed variations (scan_speed_varations.py:33-60)",
         what="the sine's amplitude doubled (stddev / 50 instead of / 100)",
         edits=[("wayne_amd/trend_generators/scan_speed_varations.py", "ssv_scaling = (self.stddev / 100.) * np.sin(",
                 "ssv_scaling = (self.stddev / 50.) * np.sin(")]),
    dict(name="py_sample_mid_points", stage="A12 sample loop (exposure_generator.py:531-579)",
         what="sub-sample mid-points a third into the sub-sample instead of half",
         edits=[("wayne_amd/exposure_generator.py", "sample_mid_points = sample_starts + (sample_durations / 2)",
                 "sample_mid_points = sample_starts + (sample_durations / 3)")]),
    dict(name="py_orbit_inclination", stage="f3 light curves: the planet's sky position (observation.py:293-357 via pylightcurve's orbit)",
         what="the projected y of the planet with sin(i) instead of cos(i)",
         tests=["tests/test_lightcurve.py"],
         edits=[("wayne_amd/lightcurve.py", "    Y = -r * np.sin(w + f) * np.cos(inc)", "    Y = -r * np.sin(w + f) * np.sin(inc)")]),
    # --- the visit level (observation.py:415-504)
    dict(name="obs_time_array", stage="f2 Observation: the sub-samples' times handed to the light curves (observation.py:440-445)",
         what="sample mid-points taken as seconds instead of milliseconds when turned into days",
         tests=["tests/test_visit_driver.py"],
         edits=[("wayne_amd/observation.py", "time_array = expstart + sample_mid_points / (86400. * 1000.)", "time_array = expstart + sample_mid_points / 86400.")]),
    dict(name="obs_shift_index", stage="f2 Observation: x / y shifts per exposure (observation.py:447-452)",
         what="the drift x_shifts * exposure NUMBER instead of * its index (one exposure's shift too many)",
         tests=["tests/test_visit_driver.py"],
         edits=[("wayne_amd/observation.py", "x_ref = self._try_index(self.x_ref, index_number) + self.x_shifts * index_number", "x_ref = self._try_index(self.x_ref, index_number) + self.x_shifts * number")]),
    dict(name="obs_trend_ignored", stage="f2 Observation: the visit trend's scale factor per exposure (observation.py:455-458)",
         what="the visit trend never applied",
         tests=["tests/test_visit_driver.py"],
         edits=[("wayne_amd/observation.py", "scale_factor = self._visit_trend.get_scale_factor(index_number) if self._visit_trend else None", "scale_factor = None")]),
    dict(name="obs_exp_start_units", stage="f2 Observation: exposure start times from the planner's minutes (observation.py:268-272)",
         what="the planner's minutes divided by 24 x 3600 instead of 24 x 60",
         tests=["tests/test_visit_driver.py"],
         edits=[("wayne_amd/observation.py", 'self.exp_start_times = self.visit_plan["exp_times"] / (24. * 60.) + self.start_JD', 'self.exp_start_times = self.visit_plan["exp_times"] / (24. * 3600.) + self.start_JD')]),
]


# Mutants of host code no GPU test is needed for (the FITS writer): `python scripts/mutation_audit.py cpu
# [name ...]` runs the CPU suite in a temporary copy of the tree.
CPU_MUTANTS = [
    dict(name="resume_any_start_time", stage="restart: a file is this visit's (observation.py:427; wayne_amd/observation.py exposure_file_is_whole)",
         what="--resume takes any whole file of the right name and mode for this visit's, whatever its start time",
         edits=[("wayne_amd/observation.py", "abs(float(p0[\"EXPSTART\"]) - (float(self.exp_start_times[number - 1]) - 2400000.5)) < 1e-7)",
                 "True)")]),
    dict(name="resume_counts_no_hdus", stage="restart: a file is whole (exposure_file_is_whole)",
         what="--resume accepts a file that lost trailing HDUs",
         edits=[("wayne_amd/observation.py", "if hdus is None or len(hdus) != 1 + 5 * self.NSAMP:", "if hdus is None:")]),
    dict(name="write_in_place", stage="f1 FITS writer: temporary name, then rename (wayne_amd/fitsio.py write_pieces)",
         what="files are written under their final name: an interrupted write leaves a partial file there",
         edits=[("wayne_amd/fitsio.py", "    part = path + PART_SUFFIX\n", "    part = path\n"),
                ("wayne_amd/fitsio.py", "    os.replace(part, path)\n", "    pass\n")]),
    dict(name="knob_read_live", stage="host: knobs frozen at context creation (include/wayne_hip.h wayne_ctx_set_knob)",
         what="wayne_exposure_upload looks at WAYNE_BATCH in the environment of the live context again",
         edits=[("wayne_amd/csrc/wayne_hip.hip", "    if (c->knobs.batch >= 0) kb = (int)std::min<long long>(std::max<long long>(c->knobs.batch, 1), kLaneBatchMax);",
                 "    if (const char* e = std::getenv(\"WAYNE_BATCH\")) kb = std::min(std::max(std::atoi(e), 1), kLaneBatchMax);")]),
    dict(name="fits_read_order", stage="f1 FITS layout (exposure.py:133-214: reads in reverse time order)",
         what="the reads written first read first",
         edits=[("wayne_amd/exposure.py", "            samp = n - 1 - i\n", "            samp = i\n")]),
    dict(name="fits_bunit", stage="f1 FITS extension header (exposure.py:186-199)",
         what="BUNIT = ELECTRONS instead of COUNTS",
         edits=[("wayne_amd/exposure.py", '("BUNIT", "COUNTS", "")', '("BUNIT", "ELECTRONS", "")')]),
    dict(name="fits_byte_order", stage="f1 FITS data (big-endian float64)",
         what="the image cube left in the host's byte order",
         edits=[("wayne_amd/exposure.py", 'cube = np.empty((n,) + arrs[0].shape, dtype=">f8")', 'cube = np.empty((n,) + arrs[0].shape, dtype="<f8")')]),
    dict(name="planner_buffer_dump", stage="f2 visit planner (visit_planner.py:76: 5.8 min per buffer dump)",
         what="a buffer dump takes 8.5 minutes",
         edits=[("wayne_amd/visit_planner.py", "time_buffer_dump = 5.8 ", "time_buffer_dump = 8.5 ")]),
    dict(name="planner_guide_star", stage="f2 visit planner (guide-star acquisition: 6 min in the first orbit, 5 after)",
         what="guide-star acquisition 5 minutes in every orbit",
         edits=[("wayne_amd/visit_planner.py", "guide_star_aq = 6.0 if orbit_n == 0 else 5.0", "guide_star_aq = 5.0")]),
    dict(name="hook_uses_visit_start", stage="f2 visit trends (visit_trends.py:44-73: the hook restarts every orbit)",
         what="the exponential hook measured from the visit's start instead of each orbit's",
         edits=[("wayne_amd/trend_generators/visit_trends.py", "        t_0[lo:hi] = time_array[lo]", "        t_0[lo:hi] = time_array[0]")]),
    # --- the CLI's YAML handling (run_visit.py:100-260)
    dict(name="cli_pre_crop", stage="f2 CLI: the planet spectrum's pre-crop (run_visit.py:152-153: 0.9 - 1.8 um)",
         what="the planet spectrum cropped to 1.0 - 1.7 um",
         edits=[("wayne_amd/run_visit.py", "tools.crop_spectrum(0.9, 1.8, wl_planet, depth_planet)", "tools.crop_spectrum(1.0, 1.7, wl_planet, depth_planet)")]),
    dict(name="cli_flux_scale", stage="f2 CLI: the stellar flux scale (run_visit.py:205)",
         what="flux_scale applied twice",
         edits=[("wayne_amd/run_visit.py", 'stellar_flux_scaled = flux_star * target["flux_scale"]', 'stellar_flux_scaled = flux_star * target["flux_scale"] * target["flux_scale"]')]),
    # --- the ORACLE by itself (which of its statements do the CPU tests pin, with no device in sight?)
    dict(name="oracle_rand_r", stage="oracle A2: glibc rand_r restated (psf_oracle.c)",
         what="the second LCG step of rand_r with increment 12346",
         edits=[("oracle/psf_oracle.c", "  s = s * 1103515245u + 12345u;\n  r = (s >> 16) & 2047u;", "  s = s * 1103515245u + 12346u;\n  r = (s >> 16) & 2047u;")]),
    dict(name="oracle_counts_chain", stage="oracle A9: the counts chain (wayne_oracle.py)",
         what="the 1e4 A / um factor 1 % high",
         edits=[("oracle/wayne_oracle.py", "        count_rate = count_rate * 1e4\n", "        count_rate = count_rate * 1.01e4\n")]),
    dict(name="oracle_flat_norm", stage="oracle A11: the flat's wavelength normalisation (wayne_oracle.py)",
         what="(wl - wmin) / wmax instead of / (wmax - wmin)",
         edits=[("oracle/wayne_oracle.py", "wl_array_norm = (wl_array - self.flat_wmin) / (self.flat_wmax - self.flat_wmin)",
                 "wl_array_norm = (wl_array - self.flat_wmin) / (self.flat_wmax)")]),
    dict(name="oracle_read_noise", stage="oracle A15: read noise (wayne_oracle.py)",
         what="read noise 14.8 / gain",
         edits=[("oracle/wayne_oracle.py", "self.read_noise = 14.1 / self.constant_gain", "self.read_noise = 14.8 / self.constant_gain")]),
]


def lib_of(name):
    return os.path.join(OUT_DIR, name + ".so")


def build_one(m):
    src = os.path.join(OUT_DIR, m["name"])
    shutil.rmtree(src, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(src, "include"))
    shutil.copytree(wb.CSRC, os.path.join(src, "wayne_amd", "csrc"))
    for rel, old, new in m["edits"]:
        p = os.path.join(src, "wayne_amd", "csrc", rel)
        s = open(p).read()
        if s.count(old) != 1:
            raise SystemExit("mutant %s: %r occurs %d times in %s" % (m["name"], old, s.count(old), rel))
        open(p, "w").write(s.replace(old, new))
    cmd = [wb.HIPCC] + wb.FLAGS + ["-o", lib_of(m["name"]), os.path.join(src, "wayne_amd", "csrc", "wayne_hip.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("mutant %s does not compile:\n%s" % (m["name"], r.stderr[-3000:]))
    shutil.rmtree(src, ignore_errors=True)
    return m["name"]


def verdict(output):
    """(killed by | None, tail) from a `pytest -x -q` output."""
    for line in output.splitlines():
        if line.startswith("FAILED ") or line.startswith("ERROR "):
            return line.split(" - ")[0].split(" ", 1)[1], line
    return None, output.strip().splitlines()[-1] if output.strip() else ""


def run_tests(lib, paths, extra=()):
    env = dict(os.environ, WAYNE_HIP_LIB=lib)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", "not negative_control", "-p", "no:cacheprovider"]
    cmd += list(extra) + list(paths)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    return verdict(r.stdout) + (time.time() - t0,)


def run_py_mutant(m, say):
    """A host-Python mutant: the tests run in a temporary copy of the tree with the edit applied."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "repo")
        shutil.copytree(ROOT, dst, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "mutants", "__pycache__",
                                                                 ".pytest_cache"))
        for rel, old, new in m["edits"]:
            p = os.path.join(dst, rel)
            s = open(p).read()
            if s.count(old) != 1:
                raise SystemExit("mutant %s: %r occurs %d times in %s" % (m["name"], old, s.count(old), rel))
            open(p, "w").write(s.replace(old, new))
        env = {k: v for k, v in os.environ.items() if k != "WAYNE_HIP_LIB"}

        def run(paths, marker):
            cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", marker, "-k", "not negative_control", "-p",
                   "no:cacheprovider"] + list(paths)
            t0 = time.time()
            r = subprocess.run(cmd, cwd=dst, env=env, capture_output=True, text=True, timeout=1500)
            return verdict(r.stdout) + (time.time() - t0,)

        k, tail, dt = run(INDEPENDENT + ["tests/test_reference_goldens.py"] + m.get("tests", []), "gpu or not gpu")
        if k is not None:
            say("%-22s | %s | KILLED by the independent set: %s (%.0f s)" % (m["name"], m["what"], k, dt))
            return
        k2, tail2, dt2 = run(["tests", "--deselect=tests/test_mutation_sites.py"] + ["--deselect=" + p_ for p_ in INDEPENDENT], "gpu")
        say("%-22s | %s | SURVIVED the independent set (%s, %.0f s); rest of the GPU suite: %s (%.0f s)" % (
            m["name"], m["what"], tail, dt, "killed by " + k2 if k2 else "SURVIVED: " + tail2, dt2))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else ""
    names = sys.argv[2:]
    todo = [m for m in MUTANTS if not names or m["name"] in names]
    todo_py = [m for m in PY_MUTANTS if not names or m["name"] in names]
    if what == "build":
        os.makedirs(OUT_DIR, exist_ok=True)
        with ThreadPoolExecutor(4) as ex:
            for n in ex.map(build_one, todo):
                print("built", n, flush=True)
    elif what == "run":
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        rest = ["tests"] + ["--deselect=" + p for p in INDEPENDENT]
        with open(REPORT, "a") as f:
            def say(s):
                print(s, flush=True)
                f.write(s + "\n")
                f.flush()
            # the unmutated library first: the very same commands must be green
            if not names:
                k, tail, dt = run_tests(wb.LIB, INDEPENDENT)
                say("%-22s | (the shipped library)                         | independent set: %s (%.0f s)" % (
                    "none", "PASSED: " + tail if k is None else "FAILED " + k, dt))
            for m in todo_py:
                run_py_mutant(m, say)
            for m in todo:
                k, tail, dt = run_tests(lib_of(m["name"]), INDEPENDENT + m.get("tests", []))
                if k is not None:
                    say("%-22s | %s | KILLED by the independent set: %s (%.0f s)" % (m["name"], m["what"], k, dt))
                    continue
                k2, tail2, dt2 = run_tests(lib_of(m["name"]), rest + ["--deselect=" + p for p in m.get("tests", [])])
                say("%-22s | %s | SURVIVED the independent set (%s, %.0f s); rest of the suite: %s (%.0f s)" % (
                    m["name"], m["what"], tail, dt, "killed by " + k2 if k2 else "SURVIVED: " + tail2, dt2))
    elif what == "cpu":
        import tempfile
        for m in [m for m in CPU_MUTANTS if not names or m["name"] in names]:
            with tempfile.TemporaryDirectory() as tmp:
                dst = os.path.join(tmp, "repo")
                shutil.copytree(ROOT, dst, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "mutants",
                                                                         "__pycache__", ".pytest_cache"))
                for rel, old, new in m["edits"]:
                    p = os.path.join(dst, rel)
                    s = open(p).read()
                    if s.count(old) != 1:
                        raise SystemExit("mutant %s: %r occurs %d times in %s" % (m["name"], old, s.count(old), rel))
                    open(p, "w").write(s.replace(old, new))
                t0 = time.time()
                r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "tests",
                                    "--deselect=tests/test_mutation_sites.py"],      # (it reads the sources the mutant edits)
                                   cwd=dst, capture_output=True, text=True, timeout=1500,
                                   env={k: v for k, v in os.environ.items() if k != "WAYNE_HIP_LIB"})
                k, tail = verdict(r.stdout)
                print("%-22s | %s | CPU suite: %s (%.0f s)" % (m["name"], m["what"], "KILLED by " + k if k else "SURVIVED: " + tail,
                                                              time.time() - t0), flush=True)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
