"""GPU: the DETERMINISTIC detector stages held to the reference's own statements -- with no oracle in between.

The oracle (oracle/wayne_oracle.py) restates `_add_read_reductions` and `_post_exposure_reductions`
(exposure_generator.py:468-515, 407-444) and the GPU path is compared with it elsewhere; but the oracle is the
builder's, and nothing the reference holds pins these rows (A13 / A15 "parity unpinned").  What can be
checked against the reference's TEXT directly are algebraic identities between runs of the device path that differ in
ONE switch, noise off, same electrons (deterministic reads, full size):

  non-linearity   detector.py:335-348   the read u solves  u (1 + c1 + c2 u + c3 u^2 + c4 u^3) = px  to |du| < 1e-3,
                                        px the same read with the switch off -- for four non-zero coefficient planes
  clip            exposure.py:82-92     read = min(max(px, -20), 78000), zero read included, before the zero read is added
  gain            detector.py:200-209   px /= 2.35 / pfl   against   px /= 2.35   (exposure_generator.py:507-511)
  initial bias    :446-466, exposure.py:94-104   every read += the (clipped, border-reset) bias frame, SUBARRAY 256 only
  reference pixels exposure.py:122-131  the 5-pixel border of every read is exactly 0

Both arithmetics: float64 reads (the exact chain) and float32 reads (the production chain of `bench.py`).
"""
import numpy as np
import pytest

import helpers
from wayne_amd import _lib, calibration, detector, grism, synthetic

pytestmark = pytest.mark.gpu

QUIET = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False,
             add_initial_bias=False)
_cache = {}


def lin_visit(name, E=None):
    """A visit over a calibration set with ALL FOUR linearity planes populated (the synthetic set has c2 alone)."""
    if "cal" not in _cache:
        cal = calibration.CalibrationSet.synthetic(11)
        rng = np.random.default_rng(4)
        cal.lin[0] = (2e-3 * rng.normal(1, 0.2, (1024, 1024))).astype(np.float32)           # c1
        cal.lin[2] = (-3e-12 * rng.normal(1, 0.2, (1024, 1024))).astype(np.float32)         # c3
        cal.lin[3] = (2e-17 * rng.normal(1, 0.2, (1024, 1024))).astype(np.float32)          # c4
        _cache["cal"] = cal
    cal = _cache["cal"]
    gr = grism.G141(cal)
    return synthetic.Visit(name, detector.WFC3_IR(), gr, cal, E=E)


def reads_of(v, out_dtype, **over):
    """Deterministic reads.  float32: a faint sky switches the table-driven sky draw on, and with it the production
    (all-float32) chain of k_ramp -- its counts come from per-pixel streams keyed by (seed, exposure, pixel), so two runs
    that differ in a detector switch hold the SAME electrons and the identities are untouched."""
    from wayne_amd import engine
    kw = dict(QUIET, **over)
    if out_dtype == np.float32:
        kw["sky_background"] = 1.0
    pg = helpers.product_generator(v, 0)
    exp = pg.scanning_frame(out_dtype=out_dtype, rng_mode=_lib.RNG_SPLIT, **v.frame_kwargs(0, **kw))
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, kw["add_initial_bias"])
    want = "k_ramp<float, true, 1, false, false>" if out_dtype == np.float32 else "k_ramp_wide<double, true, 0, false>"
    assert eng.ctx.ramp_variant(0) == want
    return np.stack([np.asarray(r[0], dtype=np.float64) for r in exp.reads])


@pytest.mark.parametrize("out_dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_non_linearity_solves_the_references_quartic(out_dtype):
    v = lin_visit("cfg4", E=4e9)                      # up to ~77 000 DN in the last read: the detector's whole range
    S = v.detector.full_size(v.SUBARRAY)
    # the coefficient planes stay float32, as the reference's are: `1 + c1`, `2 * c2` ... are then float32 sums and
    # products in numpy, and only meet the float64 pixel values afterwards (detector.py:340-341, evaluated as written)
    lin = [np.asarray(p, dtype=np.float32)[512 - S // 2:512 + S // 2, 512 - S // 2:512 + S // 2] for p in v.calibration.lin]
    px = reads_of(v, out_dtype, add_non_linear=False, clip_values_det_limits=False)
    u = reads_of(v, out_dtype, add_non_linear=True, clip_values_det_limits=False)
    assert 6e4 < px[-1].max() < 1.2e5 and not u[0].any() and not px[0].any()
    c1, c2, c3, c4 = lin
    forward = u * (1 + c1 + u * (c2 + u * (c3 + c4 * u)))                       # detector.py:340
    slope = 1 + c1 + 2 * c2 * u + 3 * c3 * u * u + 4 * c4 * u * u * u           # its derivative (:341)
    assert forward.dtype == np.float64 and (1 + c1).dtype == np.float32
    # Newton's stop |du| < 1e-3 in u is |f(u) - px| < 1e-3 f'(u) in px; float32 reads add their own rounding
    tol = 1e-3 * np.abs(slope) + (2e-7 * np.abs(px) + 0.02 if out_dtype == np.float32 else 1e-9 * np.abs(px) + 1e-9)
    err = np.abs(forward - px)
    assert (err <= tol).all(), "worst %.3g DN beyond the stop at %.1f DN" % (float((err - tol).max()), float(px.flat[np.argmax(err - tol)]))
    # and the solve did something: the response bends by thousands of DN at the top of the ramp
    assert float((px - u).max()) > 500.0
    # the iteration converged far inside the reference's stop (quadratic convergence from a warm start)
    assert float(np.median(err[px > 1000] / np.abs(slope[px > 1000]))) < (1e-6 if out_dtype == np.float64 else 5e-3)
    assert not u[:, :5, :].any() and not u[:, -5:, :].any() and not u[:, :, :5].any() and not u[:, :, -5:].any()


@pytest.mark.parametrize("out_dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_clip_is_the_references_clamp(out_dtype):
    v = lin_visit("cfg4", E=3e10)                     # the core of the spectrum saturates: > 78 000 DN without the clip
    free = reads_of(v, out_dtype, add_non_linear=False, clip_values_det_limits=False)
    clipped = reads_of(v, out_dtype, add_non_linear=False, clip_values_det_limits=True)
    assert free[-1].max() > 9e4
    want = np.clip(free, -20.0, 78000.0)
    np.testing.assert_allclose(clipped, want, rtol=0, atol=0.0 if out_dtype == np.float64 else 0.02)
    assert (clipped == 78000.0).sum() > 1000          # saturated pixels sit exactly on the limit


@pytest.mark.parametrize("out_dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_gain_variations_are_a_division_by_the_pixel_flat_gain(out_dtype):
    v = helpers.make_visit("cfg4")
    S = v.detector.full_size(v.SUBARRAY)
    flat_gain = reads_of(v, out_dtype, add_gain_variations=False, add_non_linear=False, clip_values_det_limits=False)
    var_gain = reads_of(v, out_dtype, add_gain_variations=True, add_non_linear=False, clip_values_det_limits=False)
    pfl = np.ones((S, S), dtype=np.float32)
    pfl[5:-5, 5:-5] = v.calibration.pfl                                # SUBARRAY 1024: the 1014^2 plane as it is
    gain = (np.float32(2.35) / pfl).astype(np.float64)                # float32 division, as numpy's scalar / f32 array
    want = flat_gain * 2.35 / gain                                     # same electrons: px / gain against px / 2.35
    np.testing.assert_allclose(var_gain, want, rtol=3e-7 if out_dtype == np.float32 else 1e-12,
                               atol=0.02 if out_dtype == np.float32 else 1e-9)
    assert np.abs(var_gain[-1] - flat_gain[-1]).max() > 10.0           # (the pixel flat is there: 1 % of thousands of DN)


@pytest.mark.parametrize("out_dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_dark_current_is_a_normal_about_the_super_dark_with_the_references_error_floor(out_dtype):
    # detector.py:185-191: pixel_array + np.random.normal(dark, np.where(err > 0, err, 0.00001)), the frames of read NSAMP
    # index n at HDU -5 n and -5 n + 1 of the mode's super-dark.  Two runs that differ in the dark switch alone (same
    # electrons, read noise / non-linearity / clip off): their difference IS the dark draw.  Where the error plane is zero
    # or negative (the synthetic planes sprinkle both) the draw sits within 6.8 x 1e-5 of the dark value -- found missing
    # by scripts/mutation_audit.py: a floor of 0.01 instead of 1e-5 passed every test that does not go through the oracle
    v = helpers.make_visit("cfg4")
    sci, err = v.calibration.dark_frames(v.SUBARRAY, v.SAMPSEQ, v.read_times)
    off = reads_of(v, out_dtype, add_dark=False, add_non_linear=False, clip_values_det_limits=False)
    on = reads_of(v, out_dtype, add_dark=True, add_non_linear=False, clip_values_det_limits=False)
    d = (on - off)[1:]                                                  # (15, S, S); the zero read has no dark
    assert not (on[0] - off[0]).any()
    d = d[:, 5:-5, 5:-5]
    sci, err = sci[:, 5:-5, 5:-5].astype(np.float64), err[:, 5:-5, 5:-5].astype(np.float64)
    # float32 reads round the sum (up to ~3e4 DN in the trace: ulp 2e-3; 2e-5 DN where only the sky is): keep to faint pixels
    faint = off[1:, 5:-5, 5:-5] < 100.0
    rnd = 2e-5 if out_dtype == np.float32 else 1e-12
    floor = (err <= 0) & faint
    assert (err[floor] == 0).sum() > 10000 and (err[floor] < 0).sum() > 5000
    dev = d[floor] - sci[floor]
    assert np.abs(dev).max() < 6.8e-5 + rnd, "a floor pixel %.2e DN from its dark value" % np.abs(dev).max()
    if out_dtype == np.float64:
        assert 0.9e-5 < dev.std() < 1.1e-5 and abs(dev.mean()) < 5e-5 / np.sqrt(dev.size)
    # everywhere else: a unit normal once standardised by the error plane as it is
    ok = (err > 0) & faint
    z = (d[ok] - sci[ok]) / np.sqrt(err[ok] ** 2 + rnd ** 2)
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.std() - 1.0) < (2e-3 if out_dtype == np.float32 else 1e-3)
    assert np.abs(z).max() < 6.9


@pytest.mark.parametrize("name", ["cfg4", "cfg5_g102", "cfg2"])
def test_counts_chain_is_the_references_product(name):
    # exposure_generator.py:600-628 and :678-684, as written: per wavelength bin and sub-sample
    #     counts = flux (1 - depth) x sensitivity(wl) x delta_lambda [um -> A: 1e4] x exptime [ms -> s: 1e-3] x scale_factor
    # with the sensitivity np.interp'ed from the grism's table (grism.py:116-118), delta_lambda = tools.bin_centers_to_widths
    # of the CROPPED grid (tools.py:106-128: half-gaps to both neighbours, the end bins mirroring theirs), and np.round when
    # the stellar noise is off (:627).  k_prep's counts against that product evaluated HERE, in numpy, from the visit's
    # arrays -- no oracle (the audit's mutant with the 1e4 factor 1 % high passed every oracle-free test before this one)
    v = helpers.make_visit(name)
    rec = {}
    kw = v.frame_kwargs(0, add_stellar_noise=False, cosmic_rate=None)
    helpers.product_generator(v, 0).scanning_frame(out_dtype=np.float32, record=rec, **kw)
    want, _ = helpers.reference_counts(v, kw, rec["dur"])          # (tests/helpers.py: the chain spelled out in numpy)
    got = np.asarray(rec["counts"], dtype=np.float64)
    assert got.shape == want.shape and want.max() > 100.0
    # np.round of a product evaluated in another order of operations: equal except where the product lies within an ulp
    # of a half-integer
    diff = got - np.round(want)
    assert np.abs(diff).max() <= 1.0 and (diff != 0).mean() < 1e-4, (np.abs(diff).max(), (diff != 0).mean())
    near_tie = np.abs(want - np.floor(want) - 0.5) < 1e-6 * np.maximum(want, 1.0)
    assert not (diff != 0)[~near_tie].any()
    assert abs(got.sum() - np.round(want).sum()) <= (diff != 0).sum()


@pytest.mark.parametrize("name", ["cfg4", "cfg3", "cfg5_g102", "tiny512"])
def test_bin_positions_are_the_references_trace_scan_and_frame_offset(name):
    # exposure_generator.py:258, 517-529, 591-594, 630-632, as written: at sub-sample i the star sits at
    # (x_ref + jitter, y_ref + mid_point_i x scan_speed), a bin of wavelength wl at trace.wl_to_x / wl_to_y of that star, and
    # on the frame at that position minus 507 - SUBARRAY / 2 (0 at the full array here: the reference's -5 there shifts
    # the spectrum off its own flat, HISTORY.md section 1, kept only with reference_quirks).  The trace is the product's
    # PYTHON `_SpectrumTrace` -- whose coefficients and wavelength map are held to the reference's own test values in
    # tests/test_reference_goldens.py -- evaluated here; the positions are the DEVICE's (k_prep_wl / k_prep_sub).  No oracle:
    # the audit's mutants "scan speed 1 % high" and "offset 512 - SUBARRAY / 2" were stopped only by tests that restate the host loop.
    from wayne_amd import tools
    v = helpers.make_visit(name)
    rec = {}
    kw = v.frame_kwargs(0, add_stellar_noise=False, cosmic_rate=None)
    helpers.product_generator(v, 0).scanning_frame(out_dtype=np.float32, record=rec, **kw)
    lo, hi = v.grism.wl_limits
    i0, i1 = tools.crop_spectrum_ind(lo, hi, v.wl.copy())
    wl = v.wl[i0:i1]
    sub_scale = 0 if v.SUBARRAY == 1024 else 507 - v.SUBARRAY // 2
    mid = np.asarray(v.sample_mid_points, dtype=float)                       # ms
    y_star = v.y_refs[0] + mid * (v.scan_speed / 1000.0)                     # px/s -> px/ms (:247)
    K = len(mid)
    assert rec["x"].shape == rec["y"].shape == (K, wl.size)
    # the star's x is x_ref + N(0, x_jitter) per sub-sample (:327-329): the draw is the product's own, its size is not
    jit = np.asarray(rec["x_ref"]) - v.x_refs[0]
    assert np.abs(jit).max() < 6 * v.x_jitter and (K < 20 or 0.5 * v.x_jitter < jit.std() < 1.5 * v.x_jitter)
    np.testing.assert_allclose(rec["y_ref"], y_star, rtol=0, atol=1e-9)      # (y_jitter is 1e-15 in these visits)
    if name != "tiny512":
        assert y_star[-1] - y_star[0] > 20.0                                 # (a real scan: 1 % of it is a fraction of a pixel or more)
    for k in sorted(set([0, K // 3, K - 1])):
        tr = v.grism.get_trace(float(rec["x_ref"][k]), float(y_star[k]))
        np.testing.assert_allclose(rec["x"][k], tr.wl_to_x(wl) - sub_scale, rtol=0, atol=1e-8, err_msg="x, sub-sample %d" % k)
        np.testing.assert_allclose(rec["y"][k], tr.wl_to_y(wl) - sub_scale, rtol=0, atol=1e-8, err_msg="y, sub-sample %d" % k)


def test_initial_bias_is_added_to_every_read_of_a_256_subarray():
    v = helpers.make_visit("small256")
    bias = np.asarray(v.calibration.bias_256, dtype=np.float64)
    for out_dtype, atol in ((np.float64, 1e-9), (np.float32, 0.02)):
        off = reads_of(v, out_dtype, add_initial_bias=False)
        on = reads_of(v, out_dtype, add_initial_bias=True)
        zero = np.clip(bias, -20.0, 78000.0)                           # exposure.py:82-92: the zero read is clipped too
        zero[:5, :] = zero[-5:, :] = 0.0                               # ... and its reference pixels reset (:122-131)
        zero[:, :5] = zero[:, -5:] = 0.0
        np.testing.assert_allclose(on[0], zero, rtol=0, atol=atol)
        np.testing.assert_allclose(on - off, np.broadcast_to(zero, on.shape), rtol=0, atol=atol * 2)
        assert np.abs(zero).max() > 100.0 and not off[0].any()
    # any other sub-array ignores the switch (exposure_generator.py:452-458)
    v128 = helpers.make_visit("tiny128")
    np.testing.assert_array_equal(reads_of(v128, np.float64, add_initial_bias=True),
                                  reads_of(v128, np.float64, add_initial_bias=False))


@pytest.mark.parametrize("name", ["cfg2", "stare_g102"])
def test_flat_field_is_the_references_cubic_in_the_pixel_wavelength(name):
    # G141.get_flat_field (grism.py:349-409): where a sub-sample's frame holds electrons it is multiplied by
    #   f0 + f1 t + f2 t^2 + f3 t^3,   t = (wl_px - WMIN) / (WMAX - WMIN),   wl_px = a_w d + b_w
    # with d the distance of the pixel along the trace of THAT sub-sample's star position -- `get_pixel_wl`, which the
    # reference's own test values pin (tests/test_reference_goldens.py) -- evaluated into a float32 array (np.ones_like of
    # a float32 plane).  A staring exposure has one sub-sample per read: with the pointing jitter ON every read has its
    # own star position, and accumulators(flat on) = accumulators(flat off) x that read's flat, pixel by pixel -- the
    # thrower's integers are the same in both runs.
    if name == "stare_g102":
        cal = helpers.calibration_set()
        v = synthetic.Visit("cfg2", detector.WFC3_IR(), grism.G102(cal), cal, E=2.5e7)
    else:
        v = helpers.make_visit(name)
    v.x_jitter, v.y_jitter = 0.3, 0.2                      # (pixels: every read's flat is visibly its own)
    g = v.grism
    rec_on, rec_off = {}, {}
    pg = helpers.product_generator(v, 0)
    for rec, flat in ((rec_on, True), (rec_off, False)):
        pg.scanning_frame(out_dtype=np.float64, record=rec, **v.frame_kwargs(0, add_flat=flat, **QUIET))
    on, off = rec_on["acc"], rec_off["acc"]
    assert np.array_equal(rec_on["counts"], rec_off["counts"]) and np.abs(off - np.rint(off)).max() < 1e-6
    assert v.K == on.shape[0] and np.array_equal(rec_on["read"], np.arange(v.K))       # one sub-sample per read
    assert np.ptp(rec_on["x_ref"]) > 0.2 and np.ptp(rec_on["y_ref"]) > 0.1
    cube = np.asarray(v.calibration.flat[g.name], dtype=np.float32)                    # (4, 1014, 1014)
    wmin, wmax = v.calibration.flat_wl[g.name]
    worst = 0.0
    flats = []
    for r in range(v.K):
        wl_px = g.get_pixel_wl_whole_detector(rec_on["x_ref"][r], rec_on["y_ref"][r])
        t = (wl_px - wmin) / (wmax - wmin)
        flat = (cube[0] + (cube[1] * t) + (cube[2] * (t * t)) + (cube[3] * (t * t * t))).astype(np.float32).astype(np.float64)
        flats.append(flat)
        n = off[r][5:-5, 5:-5]
        got = on[r][5:-5, 5:-5]
        hit = n > 0
        assert hit.sum() > 5000 and not got[~hit].any()
        # (accumulators are fixed point, 2^-28 e-: one rounding per tile flush that touches the pixel -- a handful)
        err = np.abs(got[hit] - n[hit] * flat[hit])
        tight = 64 * 2.0 ** -28 + 1e-12 * n[hit]
        worst = max(worst, float((err / n[hit]).max()))
        # (a last-bit difference of the fp64 wavelength can move the float32 rounding of the flat by one ulp, 6e-8, in a
        # pixel or two of a hundred thousand: allowed there, nowhere else)
        assert (err <= tight + 1.3e-7 * n[hit]).all(), (r, float(err.max()))
        assert int((err > tight).sum()) <= 2 + 1e-4 * hit.sum(), (r, int((err > tight).sum()))
    # and the reads' flats really differ from one another where the star moved
    assert max(float(np.abs(flats[0] - f).max()) for f in flats[1:]) > 1e-4


def test_order_of_the_post_ramp_stages():
    # _post_exposure_reductions (exposure_generator.py:407-444): dark current -> non-linearity -> clip -> reference
    # pixels -> + zero read -> read noise.  Two places where the ORDER shows in the numbers:
    #  (1) the dark current goes in BEFORE the non-linearity: switching it on moves a read by f'(u)^-1 x dark, not by dark
    #      (0.95 of it at 30 000 - 60 000 DN with these planes) -- its noise (err 0.02 DN) averages out over 10^4 pixels;
    #  (2) the read noise goes on AFTER the clip: saturated pixels scatter about 78 000 DN with the read noise's sigma
    #      instead of sitting on the limit.
    v = lin_visit("cfg4", E=4e9)
    S = v.detector.full_size(v.SUBARRAY)
    c = [np.asarray(p, dtype=np.float64)[512 - S // 2:512 + S // 2, 512 - S // 2:512 + S // 2] for p in v.calibration.lin]
    sci, _ = v.calibration.dark_frames(v.SUBARRAY, v.SAMPSEQ, v.read_times)
    base = reads_of(v, np.float64, add_non_linear=True, clip_values_det_limits=False)
    dark = reads_of(v, np.float64, add_non_linear=True, clip_values_det_limits=False, add_dark=True)
    u = base[-1]
    slope = 1 + c[0] + u * (2 * c[1] + u * (3 * c[2] + u * 4 * c[3]))
    sel = (u > 3e4) & (u < 6e4)
    assert sel.sum() > 3000
    moved = (dark[-1] - base[-1])[sel]
    want = (sci[-1].astype(np.float64) / slope)[sel]
    se = 0.02 / np.sqrt(sel.sum())
    assert abs(moved.mean() - want.mean()) < 6 * se + 1e-4, (float(moved.mean()), float(want.mean()))
    assert want.mean() < 0.97 * float(sci[-1][sel].mean())                      # (the two orders are 5 % apart here: 0.35 DN)
    # (2) which pixels saturate is read off a run without clip and without noise
    hot = lin_visit("cfg4", E=3e10)
    free = reads_of(hot, np.float64, add_non_linear=False, clip_values_det_limits=False)[-1]
    noisy = reads_of(hot, np.float64, add_non_linear=False, clip_values_det_limits=True, add_read_noise=True)[-1]
    top = noisy[free > 79000.0]
    assert top.size > 1000 and top.max() > 78010.0 and top.min() < 77990.0
    assert abs(top.mean() - 78000.0) < 6 * 6.0 / np.sqrt(top.size) and abs(top.std() - 14.1 / 2.35) < 0.3
