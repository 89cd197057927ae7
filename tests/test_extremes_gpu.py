"""GPU: extreme-value tests of every stochastic stage on FULL frames (cfg4 geometry: 1014^2, NSAMP 16) -- the largest
single-pixel deviation and the tail frequencies of each stage against the exact law of the reference's statement.

  sky            pixel += np.random.poisson(master_sky * bg_count)            exposure_generator.py:488-495
  dark, read     N(px + dark, max(err, 1e-5)); N(px, 14.1 / 2.35)             detector.py:185-198, exposure.py:61-80
  stellar        np.random.poisson(counts) per (bin, sub-sample)              exposure_generator.py:625-628
  thrower        electrons scattered over pixels                             pyparallel_menu.c:87-108

VERDICT r04 item 1: for three rounds one sky draw per exposure walked its search to the cap -- ~500 spurious electrons in
one pixel -- and every moment test in the suite was blind to it (tests/test_extremes_cpu.py shows the checkers used here
catch it, and that the old dispersion index does not).  The instantiations tested are the production ones: the timed
`k_ramp<float, true, 1, false, true>` (all detector switches on), its run-time-flag sibling `<..., false>` for the stages
in isolation, `k_prep_sub`'s Poisson draws, `k_lane` + `k_narrow`.  A negative control rebuilds the library with the
old, unbounded search (-DWAYNE_NEGCTL_SKY_RUNAWAY) and shows the sky test FAIL on it.

The measured figures go to gpurun_out/extremes.json (committed copy under profiles/).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import extreme_stats as xs
import helpers
from oracle import wayne_oracle as wo
from wayne_amd import _lib, calibration, detector, engine, grism, synthetic

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "extremes.json")
GAIN = 2.35
READ_SIGMA = 14.1 / 2.35
PROD_ALLON = "k_ramp<float, true, 1, false, true>"
PROD_FLAGS = "k_ramp<float, true, 1, false, false>"

STAR_OFF = dict(scale_factor=1e-9, cosmic_rate=None, add_stellar_noise=False)
ONLY_SKY = dict(STAR_OFF, add_dark=False, add_read_noise=False, add_non_linear=False, clip_values_det_limits=False,
                add_gain_variations=False, add_flat=False, add_initial_bias=False)


def report(key, **figures):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        d = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        d[key] = figures
        json.dump(d, open(REPORT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


_plain = {}


def plain_visit(n_exposures):
    """cfg4 over a calibration set whose non-linearity is the identity (c1..c4 = 0) and whose gain is the constant 2.35
    (pixel flat = 1): every detector SWITCH stays on -- the ALLON instantiation runs -- while a read of the background is
    exactly Poisson(sky) / 2.35 + N(dark, err) + N(0, 14.1 / 2.35)."""
    if "cal" not in _plain:
        cal = calibration.CalibrationSet.synthetic(11)
        cal.lin[:] = 0.0
        cal.pfl[:] = 1.0
        _plain["cal"] = cal
    cal = _plain["cal"]
    return synthetic.Visit("cfg4", detector.WFC3_IR(), grism.G141(cal), cal, n_exposures=n_exposures)


def run_exposure(v, i, want_variant, out_dtype=np.float32, exact_samplers=False, **over):
    pg = helpers.product_generator(v, i)
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    desc = pg.build_descriptor(eng, out_dtype=out_dtype, exact_samplers=exact_samplers, **v.frame_kwargs(i, **over))
    eng.ctx.upload(0, desc)
    assert eng.ctx.ramp_variant(0) == want_variant
    eng.ctx.run(0)
    return eng.ctx.download(0).astype(np.float64)


def sky_rates(v, sky_ct_s):
    """lam[r][y, x] of the interior pixels, in the float32 arithmetic of the reference (master_sky *= bg_count is an
    in-place float32 multiply, :489-493) and of the kernel."""
    sky = v.calibration.sky[v.grism.name].astype(np.float32)            # 1014 x 1014: the full array needs no crop
    dt = np.diff(np.concatenate([[0.0], v.read_times]))
    return [(sky * np.float32(sky_ct_s * d)).astype(np.float64) for d in dt]


def on_level_pixels(v):
    """Interior pixels whose master-sky value IS one of the levels of the sky plan (host_plan.h plan_sky: the l / L
    quantiles of the positive pixels, L = 15 // number of distinct read intervals): their remainder mean is exactly 0,
    e^-0 = 1 -- every threshold of the integer search saturates (k_ramp.h SkyRem::thr)."""
    sky = v.calibration.sky[v.grism.name].astype(np.float32)
    pos = np.sort(sky[sky > 0].ravel())
    dt = np.diff(np.concatenate([[0.0], v.read_times]))
    L = max(1, 15 // np.unique((5.0 * dt).astype(np.float32)).size)
    levels = np.array([pos[l * pos.size // L] for l in range(L)], dtype=np.float32)
    return np.isin(sky, levels)


def sky_only_tails(v, n_exposures, rng, first=0, on_level=None):
    t = None
    k_max = 0
    for i in range(first, first + n_exposures):
        reads = run_exposure(v, i, PROD_FLAGS, sky_background=5.0, **ONLY_SKY)
        assert not reads[0].any() and not reads[:, :5, :].any() and not reads[:, :, -5:].any()
        lam = sky_rates(v, 5.0)
        e = np.rint(reads[:, 5:-5, 5:-5] * GAIN)                           # cumulative electrons: integers
        assert np.abs(reads[:, 5:-5, 5:-5] * GAIN - e).max() < 2e-3
        for r in range(1, reads.shape[0]):
            k = e[r] - e[r - 1]
            assert k.min() >= 0
            k_max = max(k_max, float((k - lam[r - 1]).max()))
            if on_level is not None:
                on_level["dev"].append((k - lam[r - 1])[on_level["mask"]])
                on_level["var"].append(lam[r - 1][on_level["mask"]])
                if "bulk_k" in on_level and i == first:                   # (one exposure's 1.5e7 draws for the bulk test)
                    on_level["bulk_k"].append(k[::2, ::2].ravel())
                    on_level["bulk_lam"].append(lam[r - 1][::2, ::2].ravel())
            tr = xs.poisson_tails(k, lam[r - 1], rng)
            t = tr if t is None else t.merged(tr)
    return t, k_max


def test_sky_draws_largest_deviation_and_tail_frequencies():
    # production math (alias tables + integer-threshold remainder, hardware exp), nothing else switched on: every count of
    # every read interval of every pixel is an integer with a known Poisson law
    v = helpers.make_visit("cfg4", n_exposures=4)
    rng = np.random.default_rng(1)
    lvl = {"mask": on_level_pixels(v), "dev": [], "var": [], "bulk_k": [], "bulk_lam": []}
    t, k_max = sky_only_tails(v, 4, rng, on_level=lvl)
    s = xs.summary(t)
    # pixels that sit exactly ON their level: remainder mean 0, cdf = 1 from the first term -- all four integer
    # thresholds saturate and the count is the table's alone.  Thresholds of 0 instead (an unsaturated conversion of
    # 2^32) would add 4 electrons to every one of these draws
    dev, var = np.concatenate(lvl["dev"]), np.concatenate(lvl["var"])
    z_level = float(dev.sum() / np.sqrt(var.sum()))
    s["on_level_draws"], s["on_level_mean_z"] = int(dev.size), z_level
    assert dev.size >= 7 * 15 * 4 and abs(z_level) < 5.0, (dev.size, z_level, float(dev.mean()))
    chi2, p_bulk, n_bulk = xs.poisson_pit_uniformity(np.concatenate(lvl["bulk_k"]), np.concatenate(lvl["bulk_lam"]), rng)
    s["bulk_pit_chi2_49dof"], s["bulk_pit_p"], s["bulk_pit_draws"] = chi2, p_bulk, n_bulk
    report("sky/production_math", largest_excess_electrons=k_max, **s)
    assert p_bulk > 1e-6, "the bulk of the sky draws is not Poisson: chi2 = %.1f on 49 degrees of freedom (%d draws)" % (chi2, n_bulk)
    assert t.n == 4 * 15 * 1014 * 1014
    bad = xs.check(t, "sky")
    assert not bad, "; ".join(bad) + "\n%r" % s
    assert k_max < 60.0            # electrons above the mean, at rates of 14-50 per interval: the runaway was +500


SKY_VARIANTS = {
    #            sky e-/s  exact   out        the instantiation it must run
    "exact_f64": (5.0,     True,   np.float64, "k_ramp_wide<double, false, 1, false>"),
    "direct":    (20.0,    False,  np.float32, "k_ramp_wide<float, true, 0, false>"),
    "pieces":    (5.0,     False,  np.float32, "k_ramp<float, true, 2, false, false>"),
}


def hot_sky_visit(n_exposures):
    """cfg4 over a master sky with hot pixels (x 3 ... x 50 of their neighbours): such a pixel lies far above its level,
    its remainder mean is tens to thousands of electrons, and the host selects the sampler that draws the remainder in
    pieces of mean <= 16 (k_ramp.h sky_draw, PIECES; host_plan.h plan_sky)."""
    if "hot" not in _plain:
        cal = calibration.CalibrationSet.synthetic(11)
        rng = np.random.default_rng(77)
        sky = cal.sky["G141"]
        yy, xx = rng.integers(0, 1014, 40), rng.integers(0, 1014, 40)
        sky[yy, xx] *= np.repeat(np.float32([3.0, 10.0, 25.0, 50.0]), 10)
        _plain["hot"] = (cal, (yy, xx))
    cal, where = _plain["hot"]
    return synthetic.Visit("cfg4", detector.WFC3_IR(), grism.G141(cal), cal, n_exposures=n_exposures), where


@pytest.mark.parametrize("name", list(SKY_VARIANTS))
def test_sky_draws_of_the_other_samplers(name):
    # the same two questions of the sky's other samplers: the exact-math alias path with float64 reads (parity runs),
    # the DIRECT per-pixel sampler a bright sky falls back to (rates that fit no alias table: Knuth below 10, PTRS above,
    # lane-asynchronously into LDS), and the alias path whose remainder is drawn in PIECES (hot pixels of the master sky)
    rate, exact, out_dtype, variant = SKY_VARIANTS[name]
    if name == "pieces":
        v, (hy, hx) = hot_sky_visit(3)
    else:
        v, (hy, hx) = helpers.make_visit("cfg4", n_exposures=3), (None, None)
    rng = np.random.default_rng(11)
    lam = sky_rates(v, rate)
    t, k_max, hot, bulk_k, bulk_lam = None, 0.0, [], [], []
    for i in range(3):
        reads = run_exposure(v, i, variant, out_dtype=out_dtype, sky_background=rate, exact_samplers=exact, **ONLY_SKY)
        e = np.rint(reads[:, 5:-5, 5:-5] * GAIN)
        assert np.abs(reads[:, 5:-5, 5:-5] * GAIN - e).max() < 0.05        # (float32 reads of up to 4e4 electrons)
        for r in range(1, reads.shape[0]):
            k = e[r] - e[r - 1]
            assert k.min() >= 0
            k_max = max(k_max, float(((k - lam[r - 1]) / np.sqrt(lam[r - 1])).max()))
            tr = xs.poisson_tails(k, lam[r - 1], rng)
            t = tr if t is None else t.merged(tr)
            if hy is not None:
                hot.append(((k - lam[r - 1]) / np.sqrt(lam[r - 1]))[hy, hx])
            if i == 0:                                                   # (one exposure's draws, thinned, for the bulk test)
                bulk_k.append(k[::2, ::2].ravel())
                bulk_lam.append(lam[r - 1][::2, ::2].ravel())
    s = xs.summary(t)
    s["largest_excess_sigma"] = k_max
    # the body of the law between 2 and 3.5 sigma is out of the tails' sight (q <= 1e-4 starts at 3.7 sigma): the audit's
    # mutant that lets PTRS accept without its density test for 0.03 <= us < 0.07 -- + 8 % of variance at these rates --
    # left every tail count of the direct sampler where it belongs
    chi2, p_bulk, n_bulk = xs.poisson_pit_uniformity(np.concatenate(bulk_k), np.concatenate(bulk_lam), rng)
    s["bulk_pit_chi2_49dof"], s["bulk_pit_p"], s["bulk_pit_draws"] = chi2, p_bulk, n_bulk
    if hot:
        z = np.concatenate(hot)               # 40 hot pixels x 15 intervals x 3 exposures, means of 40 ... 2600 electrons
        s["hot_pixel_draws"], s["hot_pixel_z_mean"], s["hot_pixel_z_std"] = int(z.size), float(z.mean()), float(z.std())
        assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.std() - 1.0) < 5 / np.sqrt(2.0 * z.size)
    report("sky/" + name, **s)
    bad = xs.check(t, "sky " + name)
    assert not bad, "; ".join(bad) + "\n%r" % s
    assert p_bulk > 1e-6, "sky %s: the bulk of the draws is not Poisson: chi2 = %.1f on 49 degrees of freedom (%d draws)" % (name, chi2, n_bulk)


def background_law(v, sky_ct_s):
    """Per read r = 1..R of a pixel of the bordered frame: (lam_cum, dark mean, normal sigma)."""
    S = v.detector.full_size(v.SUBARRAY)
    lam = sky_rates(v, sky_ct_s)
    sci, err = v.calibration.dark_frames(v.SUBARRAY, v.SAMPSEQ, v.read_times)
    err = np.where(err > 0, err, np.float32(1e-5)).astype(np.float64)
    interior = np.zeros((S, S), dtype=bool)
    interior[5:-5, 5:-5] = True
    out, cum = [], np.zeros((S, S))
    for r in range(len(lam)):
        cum = cum.copy()
        cum[5:-5, 5:-5] += lam[r]
        mean = np.where(interior, sci[r].astype(np.float64), 0.0)
        sig = np.where(interior, np.sqrt(err[r] ** 2 + READ_SIGMA ** 2), READ_SIGMA)
        out.append((cum, mean, sig))
    return out


def test_dark_and_read_noise_normals_in_the_benchmarked_instantiation():
    # k_ramp<float, true, 1, false, true> with the star and (practically) the sky switched off: every read of every pixel
    # is an independent normal -- N(dark_r, err_r) + N(0, 14.1 / 2.35) inside, the read noise alone in the reference
    # pixels and the zero read (detector.py:185-198; exposure.py:61-68, 122-131).  Hardware log2 / sqrt / sin / cos.
    v = plain_visit(3)
    law = background_law(v, 1e-7)
    t = None
    worst = 0.0
    for i in range(3):
        reads = run_exposure(v, i, PROD_ALLON, sky_background=1e-7, **STAR_OFF)
        t0 = xs.normal_tails(reads[0], 0.0, READ_SIGMA)
        t = t0 if t is None else t.merged(t0)
        worst = max(worst, float(np.abs(reads[0]).max() / READ_SIGMA))
        for r, (_, mean, sig) in enumerate(law):
            tr = xs.normal_tails(reads[r + 1], mean, sig)
            t = t.merged(tr)
            worst = max(worst, float((np.abs(reads[r + 1] - mean) / sig).max()))
    s = xs.summary(t)
    # ... and the body of the law (hardware log2 / sqrt / sin / cos in Box-Muller): the last exposure's reads 3, 8 and 15
    rng = np.random.default_rng(3)
    pick = [3, 8, 15]
    chi2, p_bulk, n_bulk = xs.normal_pit_uniformity(np.stack([reads[r] for r in pick]),
                                                    np.stack([law[r - 1][1] for r in pick]),
                                                    np.stack([law[r - 1][2] for r in pick]), rng)
    s["bulk_pit_chi2_49dof"], s["bulk_pit_p"], s["bulk_pit_draws"] = chi2, p_bulk, n_bulk
    report("normals/allon", largest_abs_z=worst, **s)
    assert p_bulk > 1e-6, "the bulk of the reads is not normal: chi2 = %.1f on 49 degrees of freedom (%d draws)" % (chi2, n_bulk)
    assert t.n == 3 * 16 * 1024 * 1024
    bad = xs.check(t, "dark + read noise")
    assert not bad, "; ".join(bad) + "\n%r" % s
    # Box-Muller from a uniform >= 2^-33: nothing beyond 6.77 sigma of either normal, and the tail is not cut short
    assert 5.0 < worst < 7.0


def test_background_reads_of_the_benchmarked_instantiation():
    # ... and with the sky on (5 e-/s): a read is Poisson(cumulative sky) / 2.35 + the two normals; sky words and normals
    # come from ONE stream per pixel in the production layout (STAGE_READ).  Exact tails by convolution for the
    # candidates; reads of a pixel share their cumulative sky, so the frequency bands allow for that dependence
    v = plain_visit(3)
    law = background_law(v, 5.0)
    t = None
    excess = 0.0
    for i in range(3):
        reads = run_exposure(v, i, PROD_ALLON, sky_background=5.0, **STAR_OFF)
        for r, (cum, mean, sig) in enumerate(law):
            tr = xs.poisson_plus_normal_tails(reads[r + 1], cum, GAIN, mean, sig)
            t = tr if t is None else t.merged(tr)
            excess = max(excess, float((reads[r + 1] - cum / GAIN - mean).max()))
    s = xs.summary(t)
    report("background/allon", largest_excess_dn=excess, **s)
    bad = xs.check(t, "background read", dependence=4.0)
    assert not bad, "; ".join(bad) + "\n%r" % s
    assert excess < 110.0          # DN above the mean on a sigma of <= 13 DN; one runaway draw is +213 DN in every later read


def test_stellar_counts_largest_deviation_and_tail_frequencies():
    # k_prep_sub's Poisson draw per (bin, sub-sample) -- fp64 PTRS behind an fp32 squeeze from a mean of 10, inversion
    # below -- at the benchmarked size (128 x 4494 bins per exposure), against the reference's counts chain
    # (exposure_generator.py:600-628) evaluated in numpy from the visit's arrays (tests/helpers.py reference_counts: no oracle)
    v = helpers.make_visit("cfg4", n_exposures=6)
    rng = np.random.default_rng(2)
    t = None
    all_k, all_lam = [], []
    for i in range(6):
        # a dimmer star in every other exposure: the edges of the spectrum then fall below a mean of 10 (the inversion sampler)
        scale = v.scale_factor(i) * (1.0 if i % 2 == 0 else 3e-3)
        rec = {}
        pg = helpers.product_generator(v, i)
        kw = v.frame_kwargs(i, scale_factor=scale, cosmic_rate=None)
        pg.scanning_frame(out_dtype=np.float32, record=rec, **kw)
        lam, _ = helpers.reference_counts(v, kw, rec["dur"])
        assert rec["counts"].shape == lam.shape == (128, 4494)
        tr = xs.poisson_tails(rec["counts"], lam, rng)
        t = tr if t is None else t.merged(tr)
        all_k.append(rec["counts"].ravel())
        all_lam.append(lam.ravel())
    s = xs.summary(t)
    # the BULK of the law too (the audit's mutant: PTRS accepting without its density test in part of the proposal region
    # left tails and extremes alone): the randomised PIT of every draw is uniform
    chi2, p_bulk, n_bulk = xs.poisson_pit_uniformity(np.concatenate(all_k), np.concatenate(all_lam), rng)
    s["bulk_pit_chi2_49dof"], s["bulk_pit_p"], s["bulk_pit_draws"] = chi2, p_bulk, n_bulk
    report("stellar/k_prep_sub", **s)
    assert t.n == 6 * 128 * 4494
    bad = xs.check(t, "stellar counts")
    assert not bad, "; ".join(bad) + "\n%r" % s
    assert p_bulk > 1e-6, "the bulk of the stellar counts is not Poisson: chi2 = %.1f on 49 degrees of freedom (%d draws)" % (chi2, n_bulk)


def test_thrower_largest_single_pixel_deviation_at_full_size():
    # k_lane + k_narrow on the benchmarked exposure (10^9 electrons): every accumulator of every read interval against
    # the EXACT law of the reference's thrower for the counts and positions the device itself reports (flat off: integer
    # electrons).  A pixel's count is a sum of independent binomials -- per bin, component and sub-sample,
    # pyparallel_menu.c:87-108 -- whose tails come from the lattice saddlepoint formula (extreme_stats.pb_tail_saddle,
    # pinned to the exact pmf on the CPU); pixels with fewer than 20 expected electrons (cell probabilities < 3e-4: the
    # law is Poisson to that precision) take the Poisson tails.  Asked of every accumulator: the largest deviation is one
    # the law produces, the tails at 1e-4 / 1e-5 are populated as the law says (the binomial samplers of k_narrow and
    # the Box-Muller of k_lane, in production math, far from their means), and nothing lands outside +-48 px of the trace.
    v = helpers.make_visit("cfg4", n_exposures=2)
    g = v.grism
    i0, i1 = wo.crop_spectrum_ind(g.wl_limits[0], g.wl_limits[1], v.wl.copy())
    s_wl = v.wl[i0:i1]
    ratio, sl, sh = (np.polyval(p.coeffs, s_wl) for p in (g.psf_ratio_poly, g.psf_sigmal_poly, g.psf_sigmah_poly))
    rng = np.random.default_rng(3)
    t_dim = t_bright = None
    worst_excess = 0.0
    for i in range(2):
        rec = {}
        pg = helpers.product_generator(v, i)
        pg.scanning_frame(out_dtype=np.float32, record=rec, **v.frame_kwargs(i, add_flat=False, cosmic_rate=None))
        acc, counts, x, y, read_of = rec["acc"], rec["counts"].astype(np.float64), rec["x"], rec["y"], rec["read"]
        assert np.abs(acc - np.rint(acc)).max() < 1e-6
        S = acc.shape[1]
        total = 0.0
        for r in range(acc.shape[0]):
            ks = np.nonzero(read_of == r)[0]
            # window: the trace of the interval's sub-samples +- 48 px (6.9 sigma_h + a few), frame coordinates
            x0, x1 = int(np.floor(x[ks].min())) - 48, int(np.floor(x[ks].max())) + 49
            y0, y1 = int(np.floor(y[ks].min())) - 48, int(np.floor(y[ks].max())) + 49
            x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, S - 10), min(y1, S - 10)
            mean = np.zeros((y1 - y0, x1 - x0))
            second = np.zeros_like(mean)
            for k in ks:
                m_, s_ = xs.thrower_window_moments(counts[k], x[k], y[k], ratio, sl, sh, x0, x1, y0, y1)
                mean += m_
                second += s_
            got = acc[r][5 + y0:5 + y1, 5 + x0:5 + x1]
            outside = acc[r].sum() - got.sum()
            assert outside == 0.0, "read %d: %g electrons outside the +-48 px window" % (r, outside)
            total += got.sum()
            live = mean > 1e-9
            assert not got[~live].any()
            dim = live & (mean < 20.0)
            assert (second[dim] / mean[dim]).max() < 3e-3      # (the electron-weighted cell probability: Poisson to that)
            tr = xs.poisson_tails(got[dim], mean[dim], rng)
            t_dim = tr if t_dim is None else t_dim.merged(tr)
            # the bright pixels: exact tails for the candidates (|z| > 3 of the exact variance)
            by, bx = np.nonzero(mean >= 20.0)
            kb, mb = got[by, bx], mean[by, bx]
            zb = (kb - mb) / np.sqrt(mb - second[by, bx])

            def terms_of(j):
                return xs.thrower_pixel_terms(counts[ks], x[ks], y[ks], ratio, sl, sh, x0 + bx[j], y0 + by[j])

            tb = xs.poisson_binomial_tails(kb, zb, terms_of, rng)
            t_bright = tb if t_bright is None else t_bright.merged(tb)
            worst_excess = max(worst_excess, float(np.abs(zb).max()))
        assert abs(total - counts.sum()) <= 1e-6 * counts.sum()                 # (the spectrum sits well inside the frame)
    sd, sb = xs.summary(t_dim), xs.summary(t_bright)
    report("thrower/cfg4", largest_abs_z_exact_variance=worst_excess, dim_pixels=sd, bright_pixels=sb)
    bad = xs.check(t_dim, "thrower, pixels below 20 e-") + xs.check(t_bright, "thrower, pixels from 20 e-", qs=(1e-4, 1e-5))
    assert not bad, "; ".join(bad) + "\n%r\n%r" % (sd, sb)
    assert t_bright.n > 2e5 and t_dim.n > 2e5 and 4.0 < worst_excess < 6.5


@pytest.mark.parametrize("mode", [_lib.RNG_SPLIT, _lib.RNG_PHILOX], ids=["k_lane", "k_throw"])
def test_wide_electrons_populate_the_gaussian_tail_out_to_their_reach(gpu_ctx, mode):
    # pyparallel_menu.c:87-108: an electron lands at (int)(x + sigma z_x), (int)(y + sigma z_y), z a pair of Box-Muller normals --
    # a 2-D gaussian whose radius has the tail exp(-r^2 / 2).  k_lane draws the radius from ONE 16-bit half-word and
    # subdivides its last cell (R > 4.7 sigma, 1.5e-5 of the electrons) by a side stream; without that subdivision every
    # such electron would sit AT 4.855 sigma and nothing beyond -- a defect the audit (scripts/mutation_audit.py) showed
    # only same-counter parity could see.  4 x 10^8 wide electrons from one position: the pixels between 5 and 6.8 sigma
    # hold what the gaussian's exact pixel masses say (~1300, ~180, ~6 electrons in three rings), and none lies beyond
    # the draw's reach (6.87 sigma in k_lane, 6.76 in the per-electron thrower: u >= 2^-33 / 2^-32)
    from scipy.stats import norm
    B, n_each, calls, sig, cx, cy, N = 50000, 4000, 2, 5.5, 507.3, 507.6, 1014       # (a thrower call takes <= 65536 bins)
    counts = np.full(B, n_each, dtype=np.int32)
    one = np.ones(B)
    f = np.zeros((N, N))
    for e in range(calls):
        f += np.asarray(gpu_ctx.psf_apply(counts, cx * one, cy * one, 1.0 * one, 0.7 * one, sig * one, N, N, 31337, 1,
                                          rng_mode=mode, exposure=e), dtype=np.float64).reshape(N, N)
    total = float(B) * n_each * calls
    assert f.sum() == total
    edges = np.arange(N + 1, dtype=np.float64)
    px = np.diff(norm.cdf((edges - cx) / sig))                       # P(column): (int) truncation = floor on a positive position
    py = np.diff(norm.cdf((edges - cy) / sig))
    mean = total * np.outer(py, px)
    yy, xx = np.mgrid[0:N, 0:N]
    d = np.hypot(xx + 0.5 - cx, yy + 0.5 - cy) / sig                  # pixel centre, in sigma
    got, want = {}, {}
    for ring, (lo, hi) in {"5.0-5.4": (5.0, 5.4), "5.4-6.0": (5.4, 6.0), "6.0-6.8": (6.0, 6.8)}.items():
        m = (d >= lo) & (d < hi)
        got[ring], want[ring] = float(f[m].sum()), float(mean[m].sum())
    beyond = float(f[d >= 7.05].sum())
    report("thrower/far_tail/" + ("k_lane" if mode == _lib.RNG_SPLIT else "k_throw"), electrons=total, observed=got,
           expected=want, beyond_7p05_sigma=beyond)
    assert 1000 < want["5.0-5.4"] < 1700 and 100 < want["5.4-6.0"] < 260 and 2 < want["6.0-6.8"] < 12
    for ring in got:
        assert abs(got[ring] - want[ring]) < 5.0 * np.sqrt(want[ring]) + 3.0, (ring, got[ring], want[ring])
    assert beyond == 0.0
    # and the core is where it belongs: the whole frame against the exact masses, pixel by pixel
    z = (f - mean) / np.sqrt(np.maximum(mean * (1.0 - mean / total), 1e-300))
    assert np.abs(z[mean > 50.0]).max() < 6.0


@pytest.mark.parametrize("sigmas", [(0.7,), (0.7, 0.85)], ids=["pooled_rows", "own_chains"])
def test_narrow_electrons_fill_their_window_with_the_gaussians_masses(gpu_ctx, sigmas):
    # The narrow component as multinomials (k_narrow): cell masses from a fit of the gaussian tail, taken as 0 beyond
    # 6.5 sigma_l, in a window of +-6 px.  The audit's mutant that cuts the tail at 4 sigma_l (6e-5 of the electrons per
    # axis put somewhere else) passed every oracle-free test.  2 x 10^8 all-narrow electrons from one position: every
    # column and every row of the frame against the gaussian's exact masses (pyparallel_menu.c:87-108), the few hundred
    # electrons in the columns / rows wholly beyond 4 sigma_l in particular, and nothing beyond the window.  Bins that coincide pool their row chains
    # (k_narrow.h "pooled rows"); bins whose sigma_l alternate by 20 % do not (each runs its own chains): both.
    from scipy.stats import norm
    B, n_each, cx, cy, N = 50000, 4000, 507.3, 507.6, 1014
    counts = np.full(B, n_each, dtype=np.int32)
    one = np.ones(B)
    sl = np.resize(np.asarray(sigmas, dtype=float), B)
    f = np.asarray(gpu_ctx.psf_apply(counts, cx * one, cy * one, 0.0 * one, sl, 5.5 * one, N, N, 271828, 1,
                                     rng_mode=_lib.RNG_SPLIT), dtype=np.float64).reshape(N, N)
    total = float(B) * n_each
    assert f.sum() == total
    edges = np.arange(N + 1, dtype=np.float64)
    figures = {}
    for axis, c in (("columns", cx), ("rows", cy)):
        got = f.sum(axis=0 if axis == "columns" else 1)
        p = np.mean([np.diff(norm.cdf((edges - c) / s_)) for s_ in sigmas], axis=0)     # equal numbers of bins per sigma
        want = total * p
        z = (got - want) / np.sqrt(np.maximum(want * (1.0 - p), 1e-300))
        big = want > 25.0
        assert big.sum() >= 7 and np.abs(z[big]).max() < 6.0, (axis, float(np.abs(z[big]).max()))
        centre = np.arange(N) + 0.5 - c
        far = np.abs(centre) > 4.0 * min(sigmas) + 0.5                       # wholly beyond 4 sigma of the narrower gaussian
        out = np.abs(centre) > 6.0 + 1.0                                     # beyond the multinomial's window of +-6 px
        figures[axis] = dict(beyond_4_sigma=float(got[far].sum()), expected=float(want[far].sum()), outside_window=float(got[out].sum()),
                             largest_abs_z=float(np.abs(z[big]).max()))
        assert want[far].sum() > 50.0
        assert abs(got[far].sum() - want[far].sum()) < 5.0 * np.sqrt(want[far].sum()) + 3.0, (axis, figures[axis])
        assert got[out].sum() == 0.0
    report("thrower/narrow_window/" + ("pooled_rows" if len(sigmas) == 1 else "own_chains"), electrons=total, **figures)


@pytest.mark.parametrize("path,name,N,n_exp", [("k_prep_sub", "cfg5", 1014, 8), ("k_lane_fused", "cfg5", 1014, 8),
                                              ("k_lane_fused", "cfg3", 256, 24)])
def test_cosmic_ray_hits_follow_their_three_laws(path, name, N, n_exp, monkeypatch):
    # MinMaxPossionCosmicGenerator.cosmic_frame (cosmic_rays.py:70-139): per read interval Poisson(rate N^2 / 1024^2 dt)
    # hits, each with energy randint(10000, 35000) (upper bound exclusive) at a pixel randint(0, N)^2, hits on one pixel
    # adding.  Star off: the accumulators hold the hits and nothing else -- 8 exposures x 15 intervals, ~12 000 hits.
    # The hits ride in the first workgroups of k_prep_sub (the benchmarked sequence) or, on a thin exposure -- which a
    # star this dim is -- of k_lane_fused: both, the first forced with WAYNE_NO_FUSE.  And on a SUB-ARRAY (256: the rate
    # scales by 1 / 16 -- scripts/mutation_audit.py: an unscaled rate is 2 % at the full array and passed there).
    from scipy import stats
    if path == "k_prep_sub":
        _lib.set_knob_all("no_fuse", "1")
    v = helpers.make_visit(name, n_exposures=n_exp)
    rate = 11.0
    dt = np.diff(np.concatenate([[0.0], v.read_times]))
    lam = rate * (N * N) / 1024.0 ** 2 * dt
    rng = np.random.default_rng(5)
    counts, energies, xs_, ys_ = [], [], [], []
    doubles = 0
    for i in range(n_exp):
        rec = {}
        pg = helpers.product_generator(v, i)
        pg.scanning_frame(out_dtype=np.float32, record=rec, **v.frame_kwargs(i, cosmic_rate=rate, add_flat=False, **{
            k_: v_ for k_, v_ in STAR_OFF.items() if k_ != "cosmic_rate"}))
        acc = rec["acc"]
        assert not acc[:, :5, :].any() and not acc[:, :, :5].any() and not acc[:, -5:, :].any() and not acc[:, :, -5:].any()
        for r in range(acc.shape[0]):
            yy, xx = np.nonzero(acc[r])
            e = acc[r][yy, xx]
            assert np.all(e == np.rint(e)) and (e.size == 0 or e.min() >= 10000)
            two = e >= 35000                                   # two hits on one pixel add (:134-139): counted as two
            doubles += int(two.sum())
            counts.append(e.size + int(two.sum()))
            energies.append(e[~two])
            xs_.append(xx - 5)
            ys_.append(yy - 5)
    counts = np.array(counts, dtype=float)
    e = np.concatenate(energies)
    x, y = np.concatenate(xs_), np.concatenate(ys_)
    lam_all = np.tile(lam, n_exp)
    t = xs.poisson_tails(counts, lam_all, rng)
    z_total = (counts.sum() - lam_all.sum()) / np.sqrt(lam_all.sum())
    # energies: discrete uniform on 10000 .. 34999
    ks = stats.kstest(e + rng.random(e.size), stats.uniform(loc=10000, scale=25000).cdf)
    # positions: uniform on the light-sensitive N x N pixels (chi-square on an 8 x 8 grid) and both axes use all of it
    grid = np.histogram2d(y, x, bins=8, range=[[0, N], [0, N]])[0]
    chi2 = ((grid - x.size / 64.0) ** 2 / (x.size / 64.0)).sum()
    report("cosmic/%s/%s" % (name, path), hits=int(counts.sum()), expected=float(lam_all.sum()), z_total=float(z_total), doubles=doubles,
           energy_ks_p=float(ks.pvalue), energy_min=float(e.min()), energy_max=float(e.max()), position_chi2=float(chi2),
           min_u_hi=float(t.u_hi.min()) if t.u_hi.size else None, min_u_lo=float(t.u_lo.min()) if t.u_lo.size else None)
    assert abs(z_total) < 5.0 and not xs.check(t, "hits per interval", qs=())
    edge = 100 if e.size > 5000 else 400
    assert ks.pvalue > 1e-4 and e.min() <= 10000 + edge and 34999 - edge <= e.max() <= 34999
    assert chi2 < 63 + 6 * np.sqrt(2 * 63.0)
    assert x.min() == 0 and y.min() == 0 and x.max() == N - 1 and y.max() == N - 1 or (x.max() >= N - 3 and y.max() >= N - 3)
    assert doubles <= 6 + 3.0 * float((lam_all ** 2).sum()) / (2.0 * N * N)     # (two hits on one pixel: n^2 / 2 N^2 per interval)


# ---------------------------------------------------------------------------------------------------------------
# negative control: the library with the unbounded search of rounds 1-3
# ---------------------------------------------------------------------------------------------------------------
def test_negative_control_the_unbounded_sky_search_is_caught():
    # the same sky test in a child process whose libwayne_hip.so is the negative-control build: it must report draws no
    # Poisson law produces (the runaway walks to 512: several hundred electrons in a pixel whose mean is 14-50)
    from wayne_amd import build as wb
    lib = wb.build_negctl_sky()          # (hipcc -DWAYNE_NEGCTL_SKY_RUNAWAY; prebuilt by __graft_entry__.build())
    code = ("import sys, json, numpy as np\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers, extreme_stats as xs, test_extremes_gpu as t\n"
            "v = helpers.make_visit('cfg4', n_exposures=8)\n"
            "tails, k_max = t.sky_only_tails(v, 8, np.random.default_rng(1))\n"
            "print(json.dumps({'bad': xs.check(tails, 'sky'), 'k_max': k_max, 'summary': xs.summary(tails)}))\n"
            % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, WAYNE_HIP_LIB=lib, WAYNE_ALLOW_FLAGGED_LIB="1")   # a negative-control build, on purpose
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    report("sky/negative_control_unbounded_search", **d)
    assert d["bad"] and any("most extreme high draw" in b for b in d["bad"]), d
    assert d["k_max"] > 300.0, d
