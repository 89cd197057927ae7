"""GPU: whole exposures at SUBARRAY = 1024 (N = 1014, S = 1024) -- the frame of the headline metric and of
BASELINE configs 2, 4 and 5 -- against oracle/wayne_oracle.py.

The full array has its own arithmetic in the reference: the frame offset 507 - 1024/2 (exposure_generator.py
:630-632), the flat's index offset and crop (grism.py:362-363, 406-407), S capped at 1024 (detector.py:116-124),
the linearity crop 512 +- S/2 (detector.py:328-333), gain / sky crops of a 1014 plane to "1024"
(detector.py:206-207, grism.py:420-421; empty in the reference, identity here -- SURVEY.md section 7).  The
shapes are the BASELINE configurations' (cfg2 staring K = 15; cfg4 / cfg5 scans, here K = 16 sub-samples)
at E ~ 2e6 electrons so that the numpy oracle finishes in seconds.

Tolerances as in test_exposure_gpu.py: counts exact; positions 1e-9 px; accumulated electrons to the
flushes' fixed-point rounding; deterministic float64 reads 1e-4 DN; same-counter noisy reads: a counted
handful of pixels whose Poisson decision flipped on a 1-ulp libm difference.
"""
import numpy as np
import pytest

import helpers
from oracle import wayne_oracle as wo
from wayne_amd import _lib

pytestmark = pytest.mark.gpu

OFF = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False)
E_SMALL = 2e6


def both(name, K=None, staring=False, thrower="oracle", rng_mode=_lib.RNG_REPLAY, quirks=False, threads=3, **over):
    v = helpers.make_visit(name, E=E_SMALL, **({} if K is None else {"K": K}))
    assert v.SUBARRAY == 1024
    kw = v.frame_kwargs(0, **over)
    pg = helpers.product_generator(v, 0)
    det, gr, eo = wo.from_calibration(v.calibration, v.grism.name, v.NSAMP, v.SAMPSEQ, v.SUBARRAY,
                                      g102_flat_quirk=quirks)
    draws = wo.PhiloxDraws(v.seed, 0, 1014)
    rec, orec = {}, {}
    common = dict(threads=threads, record=rec)
    if staring:
        skw = {k: kw[k] for k in kw if k not in ("scan_speed", "sample_rate", "ssv_generator")}
        exp = pg.staring_frame(rng_mode=rng_mode, out_dtype=np.float64, exact_samplers=True,
                               reference_quirks=quirks, **common, **skw)
        want = eo.staring_frame(threads=threads, draws=draws, thrower=thrower, record=orec,
                                reference_quirks=quirks, **helpers.oracle_kwargs(skw))
    else:
        exp = pg.scanning_frame(rng_mode=rng_mode, out_dtype=np.float64, exact_samplers=True,
                                reference_quirks=quirks, **common, **kw)
        want = eo.scanning_frame(threads=threads, draws=draws, thrower=thrower, record=orec,
                                 reference_quirks=quirks, **helpers.oracle_kwargs(kw))
    got = np.stack([r[0] for r in exp.reads])
    want = np.stack(want)
    assert got.shape == want.shape == (16, 1024, 1024)
    return v, got, want, rec, orec


def check_deterministic(v, got, want, rec, orec):
    np.testing.assert_array_equal(rec["counts"], np.stack(orec["counts"]))          # np.round of the counts chain
    np.testing.assert_allclose(rec["x"], np.stack(orec["x"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(rec["y"], np.stack(orec["y"]), rtol=0, atol=1e-9)
    acc_o = np.stack(orec["acc"])
    assert acc_o.shape == rec["acc"].shape == (15, 1024, 1024)
    np.testing.assert_allclose(rec["acc"], acc_o, rtol=1e-13, atol=4096.0 * v.K * 2.0 ** -29)
    assert acc_o.sum() > 0.8 * rec["counts"].sum()                                   # the spectrum is on the frame
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-4)
    assert not got[:, :5, :].any() and not got[:, :, -5:].any()                       # reference pixels
    assert got[-1].max() > 10


def test_cfg2_staring_deterministic():
    v, got, want, rec, orec = both("cfg2", staring=True, **OFF)
    assert v.K == 15 and len(rec["dur"]) == 15                                      # one sample per read
    check_deterministic(v, got, want, rec, orec)
    assert rec["x"].min() > 700 and rec["x"].max() < 1000        # frame offset 0: full-frame coordinates


@pytest.mark.parametrize("name", ["cfg4", "cfg5_g102"])
def test_short_scan_deterministic(name):
    # cfg4-shaped G141 scan and the G102 scan of cfg5 (own trace, limits, sensitivity and flat cube)
    v, got, want, rec, orec = both(name, K=16, **OFF)
    check_deterministic(v, got, want, rec, orec)
    rows = [np.average(np.arange(1024), weights=a.sum(axis=1) + 1e-30) for a in rec["acc"]]
    assert np.all(np.diff(rows) > 0)                                                  # the scan moves up the frame


def noisy_check(got, want, rec, orec, frac, med=1e-4):
    d = np.abs(got - want)
    bad = int((d > 1e-3 + 1e-6 * np.abs(want)).sum())
    flipped_bins = int((rec["counts"] != np.stack(orec["counts"])).sum())
    assert flipped_bins <= 3
    # one differing stellar Poisson count re-numbers the electrons of that sub-sample in the per-electron throwers
    limit = frac * got.size if flipped_bins == 0 else 0.02 * got.size
    assert bad <= limit, "%d of %d pixels differ (%d bins with a different count)" % (bad, got.size, flipped_bins)
    assert np.median(d) < med


def test_cfg5_every_switch_on_same_counters():
    # cfg5: SSV + cosmic rays + sky + dark + non-linearity + read noise + stellar noise (+ gaussian noise),
    # bit-exact replay thrower, exact samplers, Philox-keyed noise on both sides
    v, got, want, rec, orec = both("cfg5", K=16, noise_mean=1.5, noise_std=0.4)
    assert v.ssv is not None and v.cosmic_rate == 11.0
    noisy_check(got, want, rec, orec, 1e-4)
    # cosmic rays really landed (11 /s over 143 s ~ 1500 hits of 10-35 ke-)
    assert ((got[-1] - got[0]) > 3000).sum() > 500
    # the border carries read noise only
    border = np.concatenate([got[:, :5, :].ravel(), got[:, -5:, :].ravel()])
    assert abs(border.std() - 14.1 / 2.35) < 0.05


def test_cfg5_default_split_thrower_same_counters():
    # the production default (split thrower + alias-table sky) against oracle/split_oracle.c on the same counters;
    # a moved electron shows as +-1 e- in two pixels of every later read.  First without cosmic rays, so that the
    # accumulators hold the thrower's electrons only (the cosmic-ray hits go into the same accumulators) ...
    v, got, want, rec, orec = both("cfg5", K=16, thrower="split", rng_mode=_lib.RNG_SPLIT, scale_factor=40.0,
                                   cosmic_rate=None)
    acc_o = np.stack(orec["acc"])
    total = acc_o.sum()
    assert total > 5e7 and (rec["counts"] * 0.7 > 32).mean() > 0.5          # most bins take the multinomial path
    moved = np.abs(rec["acc"] - acc_o).sum() / 2
    flipped_bins = int((rec["counts"] != np.stack(orec["counts"])).sum())
    assert flipped_bins <= 3
    if flipped_bins == 0:
        assert moved <= 5e-4 * total, "%.0f of %.0f electrons moved" % (moved, total)
    d = np.abs(got - want)
    bad = int((d > 0.05 + 1e-6 * np.abs(want)).sum())
    assert bad <= (2e-3 if flipped_bins == 0 else 0.02) * got.size
    assert np.median(d) < 5e-3
    # ... then with every switch of cfg5 on
    v, got, want, rec, orec = both("cfg5", K=16, thrower="split", rng_mode=_lib.RNG_SPLIT, scale_factor=40.0)
    flipped_bins = int((rec["counts"] != np.stack(orec["counts"])).sum())
    d = np.abs(got - want)
    bad = int((d > 0.05 + 1e-6 * np.abs(want)).sum())
    assert flipped_bins <= 3 and bad <= (2e-3 if flipped_bins == 0 else 0.02) * got.size
    assert np.median(d) < 5e-3
    assert ((got[-1] - got[0]) > 3000).sum() > 500                           # the cosmic rays are there


def test_cfg2_staring_every_switch_on():
    v, got, want, rec, orec = both("cfg2", staring=True)
    noisy_check(got, want, rec, orec, 1e-4)


def test_reference_quirks_at_1024_against_oracle():
    # the reference's own arithmetic at the full array, where it runs at all: frame offset -5 (:630) with the
    # stages whose crop_central_box(.., 1024) is empty switched off (flat, gain variations, sky)
    over = dict(OFF, add_flat=False, add_gain_variations=False)
    v, got, want, rec, orec = both("cfg4", K=16, quirks=True, **over)
    check_deterministic(v, got, want, rec, orec)
    v0, got0, want0, rec0, orec0 = both("cfg4", K=16, quirks=False, **over)
    np.testing.assert_allclose(rec["x"], rec0["x"] + 5.0, atol=1e-9)
    np.testing.assert_allclose(rec["y"], rec0["y"] + 5.0, atol=1e-9)


def test_reference_quirks_flat_offset_at_1024():
    # with the quirks kept, the flat is looked up at (y - 5, x - 5) (grism.py:362-363), the counterpart of the
    # -5 frame offset: the spectrum meets the same flat pixels as without quirks, 5 px further up / right
    over = dict(OFF, add_gain_variations=False)
    v, got, want, rec, orec = both("cfg4", K=16, quirks=True, **over)
    check_deterministic(v, got, want, rec, orec)
    v0, got0, want0, rec0, orec0 = both("cfg4", K=16, quirks=False, **over)
    a, a0 = rec["acc"].sum(axis=0), rec0["acc"].sum(axis=0)
    np.testing.assert_allclose(a[15:-5, 15:-5], a0[10:-10, 10:-10], rtol=0, atol=16 * 4096.0 * 2.0 ** -28)


def test_g102_flat_quirk_at_1024():
    v, got, want, rec, orec = both("cfg5_g102", K=16, quirks=True, **dict(OFF, add_gain_variations=False))
    check_deterministic(v, got, want, rec, orec)
