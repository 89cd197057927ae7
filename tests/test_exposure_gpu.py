"""GPU: a whole exposure through wayne_exposure_synthesize (via the
ExposureGenerator mirror) against the numpy oracle, stage by stage.

Tolerances (stated per test):
  * counts per bin: exact (integers); bin positions: 1e-9 px (fp64 both sides)
  * electrons accumulated per read interval, replay thrower + flat: the device
    keeps round(n * flat * 2^28) per tile flush, so |d| <= flushes * 2^-29 e-
  * deterministic reads (noise sources off): 1e-3 DN absolute on float32 output
    is dominated by float32 rounding of values up to 78 000 DN (ulp 0.0078):
    tolerance 0.02 DN + 2e-7 relative; float64 output: 1e-6 DN
  * stochastic reads under the same Philox counters: same tolerance for all but
    a counted handful of pixels whose Poisson draw flipped on a 1-ulp logf/expf
    difference between libm and the device.
"""
import numpy as np
import pytest

import helpers
from oracle import wayne_oracle as wo
from wayne_amd import _lib

pytestmark = pytest.mark.gpu

DET_OFF = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False)


def run_both(name, i=0, thrower="oracle", rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64, threads=3, exact=True,
             **over):
    v = helpers.make_visit(name, n_exposures=max(i + 1, 1))
    kw = v.frame_kwargs(i, **over)
    pg = helpers.product_generator(v, i)
    rec = {}
    exp = pg.scanning_frame(threads=threads, rng_mode=rng_mode, out_dtype=out_dtype, record=rec,
                            exact_samplers=exact, **kw)
    got = np.stack([r[0] for r in exp.reads])
    eo = helpers.oracle_generator(v)
    orec = {}
    draws = wo.PhiloxDraws(v.seed, i, pg.detector.light_sensitive_size(v.SUBARRAY))
    want = np.stack(eo.scanning_frame(threads=threads, draws=draws, thrower=thrower, record=orec,
                                      **helpers.oracle_kwargs(kw)))
    return v, got, want, rec, orec


@pytest.mark.parametrize("name", ["tiny", "small256"])
def test_prep_counts_and_positions(name):
    v, got, want, rec, orec = run_both(name, **DET_OFF)
    np.testing.assert_array_equal(rec["counts"], np.stack(orec["counts"]))       # np.round of the counts chain
    np.testing.assert_allclose(rec["x"], np.stack(orec["x"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(rec["y"], np.stack(orec["y"]), rtol=0, atol=1e-9)
    assert rec["counts"].sum() > 0


@pytest.mark.parametrize("name", ["tiny", "small256"])
def test_prep_counts_with_stellar_poisson(name):
    v, got, want, rec, orec = run_both(name, **dict(DET_OFF, add_stellar_noise=True))
    a, b = rec["counts"], np.stack(orec["counts"])
    # fp64 PTRS both sides: a differing draw needs a 1-ulp log() difference on a decision boundary
    assert (a != b).mean() < 1e-5
    lam_mean = b.mean()
    assert abs(a.mean() - lam_mean) < 1e-3 * lam_mean


def test_prep_counts_with_stellar_poisson_over_six_decades():
    # the device decides most PTRS trials by an fp32 squeeze (k_prep.h ptrs_squeeze) and must still land on the
    # fp64 sampler's draw: expected counts from ~1 to ~3e6 per bin in one exposure
    v, _, _, rec0, _ = run_both("tiny", **dict(DET_OFF, add_flat=False))
    base = np.maximum(rec0["counts"][min(1, len(rec0["counts"]) - 1)].astype(float), 1e-3)   # expected counts as they are
    W = base.size
    target = 10.0 ** np.random.default_rng(4).uniform(0.0, 3.5, W)
    bright = np.linspace(0, W - 1, 40).astype(int)          # (kept few: the oracle throws every electron)
    target[bright] = 10.0 ** np.linspace(4.0, 6.5, 40)
    flux = v.stellar_flux * np.where(base > 0.5, target / base, 0.0)
    v, got, want, rec, orec = run_both("tiny", **dict(DET_OFF, add_stellar_noise=True, add_flat=False, stellar_flux=flux))
    a, b = rec["counts"], np.stack(orec["counts"])
    for lo, hi in ((0, 10), (10, 100), (100, 1e3), (1e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e7)):
        assert ((b >= lo) & (b < hi)).sum() >= 5, (lo, hi)
    assert (a != b).mean() < 1e-5


@pytest.mark.parametrize("table", ["gaps", "two_points", "one_point", "dense", "off_band"])
def test_sensitivity_tables_of_any_spacing(table):
    # k_prep_wl finds a bin's interval of the sensitivity table (np.interp, grism.py:116-118) by a proportional guess
    # and bisects only where that fails: tables that are far from uniform -- clustered points with a wide gap, the
    # minimum of two points, a single point, one denser than the bins, one that covers only part of the band (clamped outside) -- must
    # give the oracle's counts exactly
    import copy
    from wayne_amd import calibration as calmod, detector, grism, synthetic
    rng = np.random.default_rng(5)
    cal = copy.copy(helpers.calibration_set())
    wl0, val0 = cal.sens["G141"]
    if table == "gaps":
        wl = np.sort(np.concatenate([rng.uniform(1.0, 1.18, 150), rng.uniform(1.52, 1.8, 7), [1.0, 1.8]]))
    elif table == "two_points":
        wl = np.array([1.05, 1.72])
    elif table == "one_point":               # np.interp over one point: that value everywhere
        wl = np.array([1.4])
    elif table == "dense":
        wl = np.sort(rng.uniform(0.9, 1.9, 20000)) ** 1.0
    else:
        wl = np.linspace(1.25, 1.45, 37) + rng.uniform(0, 2e-3, 37)
    cal.sens = dict(cal.sens, G141=(wl, np.interp(wl, wl0, val0) * (1 + 0.2 * np.sin(40 * wl))))
    gr = grism.G141(cal)
    v = synthetic.Visit("tiny", detector.WFC3_IR(), gr, cal, n_exposures=1, seed=3)
    kw = v.frame_kwargs(0, **DET_OFF)
    pg = helpers.product_generator(v, 0)
    rec = {}
    pg.scanning_frame(rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64, record=rec, threads=2, **kw)
    eo = helpers.oracle_generator(v)
    orec = {}
    eo.scanning_frame(threads=2, draws=wo.PhiloxDraws(v.seed, 0, 64), thrower="oracle", record=orec, **helpers.oracle_kwargs(kw))
    np.testing.assert_array_equal(rec["counts"], np.stack(orec["counts"]))
    assert rec["counts"].sum() > 1000


@pytest.mark.parametrize("name,flat", [("tiny", True), ("tiny", False), ("small256", True)])
def test_accumulated_electrons_replay_thrower_and_flat(name, flat):
    v, got, want, rec, orec = run_both(name, **dict(DET_OFF, add_flat=flat))
    acc_o = np.stack(orec["acc"])
    assert rec["acc"].shape == acc_o.shape
    flushes = 4096.0 * v.K   # bound on tile flushes into one pixel (workgroups per sub-sample x K)
    np.testing.assert_allclose(rec["acc"], acc_o, rtol=1e-13, atol=flushes * 2.0 ** -29)
    assert acc_o.sum() > 0.5 * rec["counts"].sum() * 0.3   # a good part of the spectrum is on the frame


@pytest.mark.parametrize("name", ["tiny", "small256"])
def test_deterministic_reads_float64(name):
    v, got, want, rec, orec = run_both(name, **DET_OFF)
    assert got.dtype == np.float64 and got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-4)
    # border pixels are reference pixels: zero without read noise
    assert not got[:, :5, :].any() and not got[:, :, -5:].any()
    assert got[-1].max() > 10


def test_deterministic_reads_float32_output():
    v, got, want, rec, orec = run_both("small256", out_dtype=np.float32, **DET_OFF)
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0.02)


def test_flags_off_paths():
    over = dict(DET_OFF, add_flat=False, add_gain_variations=False, add_non_linear=False,
                clip_values_det_limits=False, add_initial_bias=False, scale_factor=None, planet_signal=None)
    v, got, want, rec, orec = run_both("tiny", **over)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-4)
    # without any detector effect the last read is (electrons on frame) / 2.35
    total_e = np.stack(orec["acc"]).sum()
    assert abs(got[-1].sum() * 2.35 - total_e) < 1e-6 * total_e


def test_full_noise_philox_everything_on():
    over = dict(noise_mean=2.0, noise_std=0.5)
    v, got, want, rec, orec = run_both("tiny", thrower="philox", rng_mode=_lib.RNG_PHILOX, **over)
    d = np.abs(got - want)
    tol = 0.05 + 1e-6 * np.abs(want)
    bad = int((d > tol).sum())
    # hardware sin/cos/log2 in the thrower move a few electrons to the next pixel, and a
    # float32 Poisson decision may flip: both show as isolated pixels off by ~1 e-/2.35
    assert bad <= 2e-3 * got.size, "%d of %d pixels differ" % (bad, got.size)
    assert np.median(d) < 5e-3


@pytest.mark.parametrize("name", ["tiny", "small256"])
def test_default_split_thrower_against_oracle_same_counters(name):
    # the production default (WAYNE_RNG_SPLIT): accumulated electrons per read interval, flat applied,
    # against oracle/split_oracle.c driven by the same counters.  A moved electron shows as +-1 in two pixels.
    # (brighter than the fixture's star so that most bins exceed the split threshold)
    v, got, want, rec, orec = run_both(name, thrower="split", rng_mode=_lib.RNG_SPLIT, **dict(DET_OFF, scale_factor=25.0))
    np.testing.assert_array_equal(rec["counts"], np.stack(orec["counts"]))
    assert (rec["counts"] * 0.7 > 32).mean() > 0.5
    acc_o = np.stack(orec["acc"])
    total = acc_o.sum()
    assert total > 1e5
    assert abs(rec["acc"].sum() - total) <= 3 + 2e-5 * total
    moved = np.abs(rec["acc"] - acc_o).sum() / 2
    assert moved <= 5e-4 * total, "%.0f of %.0f electrons moved" % (moved, total)
    # and the reads built from them: a moved electron touches two pixels of every later read
    d = np.abs(got - want)
    assert (d > 0.05 + 1e-6 * np.abs(want)).sum() <= 2 * got.shape[0] * max(moved, 1)
    assert np.median(d) < 1e-6


def test_full_noise_default_mode_everything_on():
    over = dict(noise_mean=2.0, noise_std=0.5)
    v, got, want, rec, orec = run_both("tiny", thrower="split", rng_mode=_lib.RNG_SPLIT, **over)
    d = np.abs(got - want)
    bad = int((d > 0.05 + 1e-6 * np.abs(want)).sum())
    assert bad <= 2e-3 * got.size, "%d of %d pixels differ" % (bad, got.size)
    assert np.median(d) < 5e-3


def test_noise_stages_exact_samplers_replay_thrower():
    # every noise source on, but the bit-exact replay thrower: what differs from the oracle can
    # only come from the Poisson / normal stages (same streams, libm vs ocml within 1 ulp)
    v, got, want, rec, orec = run_both("small256", **dict(add_stellar_noise=True, noise_mean=1.0, noise_std=0.3))
    d = np.abs(got - want)
    bad = int((d > 1e-3 + 1e-6 * np.abs(want)).sum())
    assert bad <= 1e-4 * got.size, "%d of %d pixels differ" % (bad, got.size)
    assert np.median(d) < 1e-4


@pytest.mark.parametrize("sky", [0.02, 3.0, 400.0])
def test_sky_levels_faint_to_bright(sky):
    # faint and ordinary skies are drawn through the shared alias tables + per-pixel remainder; a sky
    # whose rate does not fit a 256-entry table (400 e-/s) takes the direct Poisson sampler.  Same
    # streams as the oracle in every case; only the sky stage is on.
    over = dict(DET_OFF, sky_background=sky, add_non_linear=False, clip_values_det_limits=False)
    v, got, want, rec, orec = run_both("small256", **over)
    d = np.abs(got - want)
    bad = int((d > 1e-3 + 1e-6 * np.abs(want)).sum())
    assert bad <= 2e-4 * got.size, "%d of %d pixels differ" % (bad, got.size)
    # and it is a Poisson sky: mean and variance of the last read's sky away from the spectrum
    N = 256
    acc = np.stack(orec["acc"]).sum(axis=0)[5:-5, 5:-5]
    last = got[-1][5:-5, 5:-5] - got[0][5:-5, 5:-5]
    dark = acc == 0
    assert dark.sum() > 0.5 * N * N
    lam = sky * float(pgen_exptime(v))
    e = last[dark] * 2.35
    assert abs(e.mean() - lam) < 0.03 * lam + 0.02


def test_sky_plane_with_hot_dead_and_negative_pixels():
    # a master sky as a real calibration file may hold it: dead (0) and negative pixels draw nothing, hot
    # pixels (x6, x40) sit far above the top sky level, so their remainder is drawn in several pieces
    from wayne_amd import calibration, detector, grism, synthetic
    cal = calibration.CalibrationSet.synthetic(11)
    sky = cal.sky["G141"]
    c = (1014 - 256) // 2                       # inside the 256 sub-array's central crop
    sky[c + 10, c + 20] = 0.0
    sky[c + 11, c + 21] = -0.4
    sky[c + 30, c + 40] *= 6.0
    sky[c + 31, c + 41] *= 40.0
    det = detector.WFC3_IR()
    v = synthetic.Visit("small256", det, grism.G141(cal), cal, n_exposures=1)
    over = dict(DET_OFF, sky_background=2.0, add_non_linear=False, clip_values_det_limits=False,
                add_gain_variations=False)
    kw = v.frame_kwargs(0, **over)
    pg = helpers.product_generator(v, 0)
    got = np.stack([r[0] for r in pg.scanning_frame(threads=3, rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                    exact_samplers=True, **kw).reads])
    eo = helpers.oracle_generator(v)
    want = np.stack(eo.scanning_frame(threads=3, draws=wo.PhiloxDraws(v.seed, 0, 256), thrower="oracle",
                                      **helpers.oracle_kwargs(kw)))
    d = np.abs(got - want)
    assert int((d > 1e-3 + 1e-6 * np.abs(want)).sum()) <= 2e-4 * got.size
    last = (got[-1] - got[0])[5:-5, 5:-5] * 2.35
    assert last[10, 20] == 0 and last[11, 21] == 0
    lam = 2.0 * float(pgen_exptime(v))
    for (yy, xx, f) in ((30, 40, 6.0), (31, 41, 40.0)):
        mean = lam * f * float(sky[c + yy, c + xx] / f)
        assert abs(last[yy, xx] - mean) < 6 * np.sqrt(mean)
    from wayne_amd import engine
    engine.close_all()


def pgen_exptime(v):
    from wayne_amd import detector
    return detector.WFC3_IR().get_read_times(v.NSAMP, v.SUBARRAY, v.SAMPSEQ)[-1]


def test_fast_samplers_agree_with_exact():
    # production math (v_rcp/v_sqrt/v_log/v_exp/v_sin/v_cos) against the exact policy, same streams
    v = helpers.make_visit("small256")
    kw = v.frame_kwargs(0)
    pg = helpers.product_generator(v, 0)
    a = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float64, exact_samplers=True, **kw).reads])
    b = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float64, exact_samplers=False, **kw).reads])
    d = np.abs(a - b)
    # normals differ by the hardware sin/cos error (< 1e-4 sigma); a Poisson draw flips on a ~1e-6 boundary
    bad = int((d > 5e-3).sum())
    assert bad <= 1e-3 * a.size, "%d of %d pixels differ" % (bad, a.size)
    assert np.median(d) < 1e-3 and d.max() < 40.0


def test_noise_statistics_small256():
    over = dict(add_stellar_noise=True)
    v = helpers.make_visit("small256")
    kw = v.frame_kwargs(0, **over)
    pg = helpers.product_generator(v, 0)
    exp = pg.scanning_frame(out_dtype=np.float64, **kw)
    reads = np.stack([r[0] for r in exp.reads])
    # reference pixels: pure read noise N(0, 14.1/2.35) on every read (exposure.py:61-68, 122-131)
    border = np.concatenate([reads[:, :5, :].ravel(), reads[:, -5:, :].ravel()])
    # (10640 border samples of sigma 6: the mean scatters by 0.058)
    assert abs(border.mean()) < 0.25 and abs(border.std() - 14.1 / 2.35) < 0.15
    # zero read interior = clipped bias + read noise
    assert abs(np.median(reads[0][5:-5, 5:-5]) - np.median(np.clip(v.calibration.bias_256, -20, 78000)[5:-5, 5:-5])) < 2.0
    # sky-only corner far from the spectrum: mean of the first read = sky*dt/gain + dark, in DN
    dt = v.read_times[0]
    corner = reads[1][200:250, 10:60] - reads[0][200:250, 10:60]
    want = v.sky[0] * dt / 2.35 + 0.05 * dt
    assert abs(corner.mean() - want) < 0.5 + 0.05 * want


def test_exposure_is_deterministic_and_split_invariant():
    v = helpers.make_visit("small256")
    kw = v.frame_kwargs(0)
    pg = helpers.product_generator(v, 0)
    a = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    b = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    np.testing.assert_array_equal(a, b)
    import os
    _lib.set_knob_all("throw_wgs", 37)
    _lib.set_knob_all("tile_ints", 2000)
    try:
        c = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    finally:
        _lib.reset_knobs_all()
    np.testing.assert_array_equal(a, c)      # integer accumulation: launch geometry cannot change a bit
    other = helpers.product_generator(v, 1)
    d = np.stack([r[0] for r in other.scanning_frame(**kw).reads])
    assert np.abs(a - d).max() > 1.0          # a different exposure index draws different noise


def test_staring_frame_matches_scanning_at_zero_speed():
    v = helpers.make_visit("tiny")
    kw = v.frame_kwargs(0, **DET_OFF)
    pg = helpers.product_generator(v, 0)
    eg_kw = {k: kw[k] for k in kw if k not in ("scan_speed", "sample_rate", "ssv_generator")}
    st = np.stack([r[0] for r in pg.staring_frame(rng_mode=_lib.RNG_REPLAY, **eg_kw).reads])
    sc = np.stack([r[0] for r in pg.scanning_frame(rng_mode=_lib.RNG_REPLAY, **dict(kw, scan_speed=0.0)).reads])
    np.testing.assert_array_equal(st, sc)
