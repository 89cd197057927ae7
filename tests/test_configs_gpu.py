"""GPU: the BASELINE.json configurations at FULL size (1014x1014 / 256x256 frames,
NSAMP 15-16, 1e7-1e9 electrons), checked through size-independent properties --
the properties that need no oracle (whole exposures against the oracle at these
sizes: tests/test_fullsize_oracle_gpu.py):

  * conservation: with flat / gain variations off and no noise, electrons that
    reach the accumulators = electrons thrown - those falling off the frame, and
    the last read * 2.35 equals their sum;
  * the cumulative reads never decrease (noise off);
  * determinism and independence of launch geometry (bit-exact);
  * linearity of the expected counts in scale_factor;
  * statistics of the noisy production run (read noise on reference pixels, sky level).
"""
import numpy as np
import pytest

import helpers
from wayne_amd import _lib

pytestmark = pytest.mark.gpu

QUIET = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False,
             add_flat=False, add_gain_variations=False, add_non_linear=False, clip_values_det_limits=False,
             add_initial_bias=False)


def frames(v, i=0, record=None, **over):
    pg = helpers.product_generator(v, i)
    exp = pg.scanning_frame(out_dtype=np.float64, record=record, **v.frame_kwargs(i, **over))
    return np.stack([r[0] for r in exp.reads])


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg5_g102"])
def test_conservation_and_monotone_ramp(name):
    v = helpers.make_visit(name)
    rec = {}
    reads = frames(v, record=rec, **QUIET)
    thrown = int(rec["counts"].astype(np.int64).sum())
    on_frame = rec["acc"].sum()
    if name == "cfg1":
        assert v.K == 2233 and rec["counts"].shape == (2233, 4494)   # 10 ms sampling of the example visit
    # the flux scaling hits the configured E (cfg1: ~2.5 e- per bin per 10 ms sub-sample, and np.round
    # of such small expectations (exposure_generator.py:628) does not conserve the sum to better than a few %)
    assert abs(thrown - v.E) < (0.05 if name == "cfg1" else 0.02) * v.E
    lost = thrown - on_frame
    assert 0 <= lost < 0.02 * thrown                            # only PSF wings leave the frame
    assert abs(on_frame - round(on_frame)) < 1e-3               # integer electrons (no flat): exact accumulation
    np.testing.assert_allclose(reads[-1].sum() * 2.35, on_frame, rtol=1e-12)
    assert reads.shape == (v.NSAMP, v.detector.full_size(v.SUBARRAY), v.detector.full_size(v.SUBARRAY))
    assert not reads[0].any()                                    # zero read without bias / noise
    assert np.all(np.diff(reads, axis=0) >= -1e-9)               # up-the-ramp: cumulative
    # every read interval received its sub-samples' electrons
    per_read = rec["acc"].reshape(len(v.read_times), -1).sum(axis=1)
    assert np.all(per_read > 0)
    if v.scan_speed > 0:                                          # the scan moves the spectrum up the frame
        rows = [np.average(np.arange(a.shape[0]), weights=a.sum(axis=1) + 1e-30) for a in rec["acc"]]
        assert np.all(np.diff(rows) > 0)


def test_cfg4_deterministic_and_geometry_invariant():
    import os
    v = helpers.make_visit("cfg4")
    a = frames(v)
    b = frames(v)
    np.testing.assert_array_equal(a, b)
    _lib.set_knob_all("throw_wgs", 700)
    _lib.set_knob_all("tile_ints", 5000)
    try:
        c = frames(v)
    finally:
        _lib.reset_knobs_all()
    np.testing.assert_array_equal(a, c)


def test_cfg3_linearity_in_scale_factor():
    v = helpers.make_visit("cfg3")
    r1, r2 = {}, {}
    frames(v, record=r1, **dict(QUIET, scale_factor=1.0, planet_signal=None))
    frames(v, record=r2, **dict(QUIET, scale_factor=2.0, planet_signal=None))
    c1, c2 = r1["counts"].astype(np.int64), r2["counts"].astype(np.int64)
    assert np.abs(c2 - 2 * c1).max() <= 1                        # round(2 lam) vs 2 round(lam)
    # the frame doubles up to Monte-Carlo noise of the thrower
    a1, a2 = r1["acc"].sum(axis=0), r2["acc"].sum(axis=0)
    bright = a1 > 2000
    assert bright.sum() > 1000
    assert abs(np.median(a2[bright] / a1[bright]) - 2.0) < 0.01


def test_cfg4_production_statistics():
    v = helpers.make_visit("cfg4")
    reads = frames(v)
    border = np.concatenate([reads[:, :5, :].ravel(), reads[:, -5:, :].ravel()])
    assert abs(border.mean()) < 0.05 and abs(border.std() - 14.1 / 2.35) < 0.03    # reference pixels: read noise only
    # a corner the scan never reaches: sky + dark, linear in time
    t = v.read_times
    corner = reads[1:, 900:1000, 20:400].mean(axis=(1, 2))
    rate = np.polyfit(t, corner, 1)[0]
    assert abs(rate - (v.sky[0] / 2.35 + 0.05)) < 0.08                             # DN/s: sky/gain + synthetic dark
    assert reads[-1].max() > 2000 and np.isfinite(reads).all()


def test_full_frame_sky_is_poisson_in_production_mode():
    # production samplers (hardware exp / rcp, alias tables + remainder): the sky of a full 1014^2 frame,
    # nothing else switched on, must have the Poisson mean AND variance pixel by pixel.  The dispersion
    # index sum((k - lam)^2 / lam) over ~10^6 pixels is chi-square with that many degrees of freedom.
    v = helpers.make_visit("cfg4")
    over = dict(add_stellar_noise=False, cosmic_rate=None, add_dark=False, add_read_noise=False,
                add_non_linear=False, clip_values_det_limits=False, add_gain_variations=False,
                add_flat=False, add_initial_bias=False, sky_background=5.0, scale_factor=1e-9)   # star switched off
    reads = frames(v, **over)
    cal = helpers.calibration_set()
    sky = cal.sky["G141"].astype(np.float64)                       # 1014 x 1014 for SUBARRAY 1024
    dt = np.diff(np.concatenate([[0.0], v.read_times]))
    for r in (1, 2, 8, 15):
        k = (reads[r][5:-5, 5:-5] - reads[r - 1][5:-5, 5:-5]) * 2.35      # electrons of read interval r
        assert np.abs(k - np.round(k)).max() < 1e-6 and k.min() >= 0
        lam = sky * 5.0 * dt[r - 1]
        n = lam.size
        assert abs((k - lam).sum()) < 5 * np.sqrt(lam.sum())
        disp = ((k - lam) ** 2 / lam).sum()
        assert abs(disp - n) < 6 * np.sqrt(2.0 * n), (r, disp / n)
        # third moment too: skewness of a Poisson variable is lam^-1/2
        skew = (((k - lam) / np.sqrt(lam)) ** 3).mean()
        assert abs(skew - (1 / np.sqrt(lam)).mean()) < 6 * np.sqrt(15.0 / n)


def test_cfg5_cosmic_rays_and_ssv():
    v = helpers.make_visit("cfg5")
    rec = {}
    frames(v, record=rec, **dict(QUIET, cosmic_rate=11.0))
    # cosmic hits are the only thing in a region the spectrum never touches
    acc = rec["acc"]
    quiet = acc[:, 850:1000, 5:300]
    hits = quiet[quiet > 0]
    assert hits.size > 20 and hits.min() >= 10000 and np.all(hits == np.round(hits))
    expect = 11.0 * (150 * 295) / 1024 ** 2 * v.read_times[-1]
    assert 0.5 * expect < hits.size < 1.7 * expect
    # SSVSine modulates the sub-sample durations by +-1.5 %
    assert abs(rec["dur"].sum() - v.sample_durations.sum()) < 0.01 * v.sample_durations.sum()
    ratio = rec["dur"] / v.sample_durations
    assert 0.984 < ratio.min() < 0.99 and 1.01 < ratio.max() < 1.016
