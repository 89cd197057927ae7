"""CPU: the ORACLE's own statements of the rows no reference fixture pins (A9 counts chain, A11 flat, A13 / A15
detector stages) held to the reference's TEXT -- the same identities tests/test_detector_laws_gpu.py holds the device to,
here with no device in sight.

The thrower's restatement is pinned by the reference's golden frames (tests/test_psf_oracle.py); trace, mode tables, bin
widths, visit ramp by the reference's own test values (tests/test_reference_goldens.py).  For the rest the oracle is a
numpy restatement that only a line-by-line reading vouched for: scripts/mutation_audit.py (`cpu oracle_*`) planted a 1 %
error in its counts chain, a wrong normalisation in its flat and a wrong read noise, and the whole CPU suite stayed green.
Each expectation below is evaluated in the test from the formula as the reference writes it, on other code paths than
the oracle's (the pixel wavelengths come from `get_pixel_wl`, which the reference's test values pin).
"""
import numpy as np
import pytest

import helpers
from oracle import wayne_oracle as wo


@pytest.mark.parametrize("name", ["cfg3", "tiny_g102"])
def test_counts_chain_of_the_oracle_is_the_references_product(name):
    # exposure_generator.py:600-628, 678-684 (tests/helpers.py reference_counts spells the chain out in numpy)
    v = helpers.make_visit(name)
    eo = helpers.oracle_generator(v)
    kw = v.frame_kwargs(0, cosmic_rate=None)
    dur = np.asarray(v.sample_durations, dtype=float)
    want, s_wl = helpers.reference_counts(v, kw, dur)
    i0, i1 = wo.crop_spectrum_ind(eo.grism.wl_limits[0], eo.grism.wl_limits[1], v.wl.copy())
    assert i1 - i0 == s_wl.size
    eo.grism.set_current_wavelength_only_dependent_array(s_wl)
    for k in sorted(set([0, len(dur) // 2, len(dur) - 1])):
        flux = kw["stellar_flux"][i0:i1] * (1.0 - np.asarray(kw["planet_signal"])[k][i0:i1])     # combine_planet_stellar_spectrum (:688-709)
        got = eo.counts_before_noise(s_wl, flux, dur[k], kw["scale_factor"])
        np.testing.assert_allclose(got, want[k], rtol=1e-12, atol=0)
    assert want.max() > 10.0


@pytest.mark.parametrize("name,size", [("cfg3", 256), ("tiny_g102", 64), ("cfg2", 1024)])
def test_flat_field_of_the_oracle_is_the_references_cubic(name, size):
    # grism.py:349-385: flat[pixel] = f0 + f1 t + f2 t^2 + f3 t^3, t = (wl(pixel) - WMIN) / (WMAX - WMIN), wl(pixel) the
    # wavelength get_pixel_wl gives that pixel for a source at (x_ref, y_ref) (:137-163; pinned by the reference's test
    # values); on a sub-array the frame pixel (y, x) takes the flat of (y + off, x + off), off = (1014 - size) / 2 (:362-363; 0
    # at the full array here, HISTORY.md section 1)
    v = helpers.make_visit(name)
    eo = helpers.oracle_generator(v)
    gr = eo.grism
    N = min(size, 1014)
    x_ref, y_ref = float(v.x_refs[0]), float(v.y_refs[0])
    yy, xx = np.mgrid[0:N, 0:N]
    got = gr.get_flat_field(x_ref, y_ref, size=size, indices=(yy.ravel(), xx.ravel()))
    assert got.shape == (N, N) and got.dtype == np.float32
    off = 0 if size > 1014 else (1014 - size) // 2
    f0, f1, f2, f3 = (np.asarray(p, dtype=np.float32)[off:off + N, off:off + N] for p in gr.flat)
    # the flat cube lives on the 1014^2 light-sensitive array: its pixel (Y, X) sits at detector position (Y, X) in the
    # coordinates of (x_ref, y_ref) (flat_xs / flat_ys are np.meshgrid(arange(1014), arange(1014)), grism.py:74-77)
    wl_pix = gr.get_pixel_wl(x_ref, y_ref, (xx + off).astype(float), (yy + off).astype(float))
    t = (wl_pix - gr.flat_wmin) / (gr.flat_wmax - gr.flat_wmin)
    want = f0 + f1 * t + f2 * t * t + f3 * t * t * t
    np.testing.assert_allclose(got, want.astype(np.float32), rtol=3e-7, atol=0)
    assert np.ptp(want) > 1e-3 and float(np.abs(f1 * t).max()) > 1e-3 and float(np.abs(f3 * t * t * t).max()) > 1e-6   # (a cubic, not a flat of ones)


def test_detector_constants_and_stages_of_the_oracle():
    # detector.py:26-33: min / max counts -20 / 78000, gain 2.35, read noise 14.1 / gain; :185-191: err <= 0 -> 1e-5, the frames
    # of read NSAMP index n at HDU -5 n (error: -5 n + 1); :200-209: gain / pixel flat in float32; :318-350: the quartic
    v = helpers.make_visit("cfg3")
    eo = helpers.oracle_generator(v)
    det = eo.detector
    assert (det.min_counts, det.max_counts, det.constant_gain) == (-20, 78000, 2.35)
    assert det.read_noise == 14.1 / 2.35
    for n in (1, 7, 14):
        sci, err = det.dark_for_read(n)
        raw_sci, raw_err = np.asarray(det.dark_hdus[-5 * n]), np.asarray(det.dark_hdus[-5 * n + 1])
        np.testing.assert_array_equal(sci, raw_sci)
        assert (raw_err <= 0).sum() > 10
        np.testing.assert_array_equal(err, np.where(raw_err > 0, raw_err, np.float32(0.00001)))
        assert err.dtype == np.float32 and err.min() == np.float32(0.00001)
    g = det.get_gain(256)
    pfl = np.asarray(det.pfl, dtype=np.float32)
    c = (pfl.shape[0] - 256) // 2
    np.testing.assert_array_equal(g, (np.float32(2.35) / pfl)[c:c + 256, c:c + 256])
    # non-linearity: the returned u solves u (1 + c1 + c2 u + c3 u^2 + c4 u^3) = px to the reference's stop |du| < 1e-3
    rng = np.random.default_rng(3)
    S = 266
    px = rng.uniform(0.0, 7.5e4, (S, S))
    u = det.apply_non_linearity(px.copy())
    lo = len(det.lin[0]) // 2 - S // 2
    c1, c2, c3, c4 = (np.asarray(p, dtype=np.float32)[lo:lo + S, lo:lo + S] for p in det.lin)
    forward = u * (1 + c1 + u * (c2 + u * (c3 + c4 * u)))
    slope = 1 + c1 + 2 * c2 * u + 3 * c3 * u * u + 4 * c4 * u * u * u
    assert (np.abs(forward - px) <= 1e-3 * np.abs(slope) + 1e-9).all() and float((px - u).max()) > 100.0


def test_read_and_dark_noise_of_the_oracle_follow_the_references_normals():
    # detector.py:191, 198 through the oracle's draw objects: N(dark, err) and N(pixel, 14.1 / 2.35) -- with numpy's legacy
    # stream (the reference's) and with the counter-keyed streams the device mirrors
    v = helpers.make_visit("tiny128")
    eo = helpers.oracle_generator(v)
    S = 138
    for draws in (wo.LegacyDraws(11), wo.PhiloxDraws(11, 0, 128)):
        x = np.full((S, S), 100.0)
        z = (draws.read_normal(x, eo.detector.read_noise, 0) - 100.0) / (14.1 / 2.35)
        assert abs(z.std() - 1.0) < 5 / np.sqrt(2.0 * z.size) and abs(z.mean()) < 5 / np.sqrt(z.size)
        dark = np.full((S, S), 3.0)
        err = np.full((S, S), 0.02, dtype=np.float32)
        zd = (draws.dark_normal(dark, err, 0) - 3.0) / 0.02
        assert abs(zd.std() - 1.0) < 5 / np.sqrt(2.0 * zd.size) and abs(zd.mean()) < 5 / np.sqrt(zd.size)
