"""Light curves.  pylightcurve is not installed, so the checker is
oracle/lc_oracle.c -- an independent 2-D integration of the Claret disk over the planet's area --
pinned here to closed forms (uniform disk, the r^2 law, the small-planet limit of the quadratic law);
the product's host model (wayne_amd/lightcurve.py) and the device kernel (k_lightcurve) are then
compared with IT."""
import numpy as np
import pytest

from oracle import clib
from wayne_amd import lightcurve as lc

LD = [0.800627, -0.757066, 0.897268, -0.384804]      # examples/...parameters.yml:26


def lens_area(z, p):
    """Area common to the unit disk and a disk of radius p at distance z (closed form)."""
    z, p = np.broadcast_arrays(np.asarray(z, dtype=float), np.asarray(p, dtype=float))
    out = np.zeros(z.shape)
    inside = z <= 1 - p
    out[inside] = np.pi * p[inside] ** 2
    part = (z > 1 - p) & (z < 1 + p)
    zz, pp = z[part], p[part]
    k0 = np.arccos(np.clip((pp ** 2 + zz ** 2 - 1) / (2 * pp * zz), -1, 1))
    k1 = np.arccos(np.clip((1 - pp ** 2 + zz ** 2) / (2 * zz), -1, 1))
    out[part] = pp ** 2 * k0 + k1 - 0.5 * np.sqrt(np.maximum(4 * zz ** 2 - (1 + zz ** 2 - pp ** 2) ** 2, 0.0))
    return out


def test_oracle_uniform_disk_closed_form():
    z = np.linspace(0, 1.4, 141) + 0.0037          # (off the exact contacts, where the arccos form below loses digits)
    p = np.array([0.05, 0.12, 0.3])
    got = clib.lc_deficit(z, p, [0, 0, 0, 0])
    np.testing.assert_allclose(got, lens_area(z[:, None], p[None, :]) / np.pi, rtol=0, atol=1e-13)
    assert got[-1].max() == 0.0 and got[0, 1] == pytest.approx(0.0144, abs=1e-15)
    # at the contacts themselves: p^2 just inside, 0 just outside
    for pp in p:
        c = clib.lc_deficit(np.array([1 - pp, np.nextafter(1 - pp, 2), 1 + pp]), np.array([pp]), [0, 0, 0, 0])[:, 0]
        np.testing.assert_allclose(c, [pp * pp, pp * pp, 0.0], rtol=0, atol=1e-13)


def test_oracle_r2_law_closed_form():
    # I = 1 - a4 (1 - mu^2) = 1 - a4 r^2: over a planet wholly inside the disk, int r^2 dA = pi p^2 (z^2 + p^2 / 2)
    a4 = 0.6
    z = np.array([0.0, 0.2, 0.5, 0.85])
    p = np.array([0.05, 0.1, 0.15])
    got = clib.lc_deficit(z, p, [0, 0, 0, a4])
    zz, pp = z[:, None], p[None, :]
    want = (pp ** 2 - a4 * pp ** 2 * (zz ** 2 + pp ** 2 / 2)) / (1 - a4 / 2)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)


def test_oracle_quadratic_law_small_planet_limit():
    # quadratic law I = 1 - u1 (1 - mu) - u2 (1 - mu)^2 is the Claret law with a2 = u1 + 2 u2, a4 = -u2;
    # a planet much smaller than the star blocks p^2 I(z) / (1 - u1/3 - u2/6), to O(p^2) relative
    u1, u2 = 0.4, 0.25
    ld = [0.0, u1 + 2 * u2, 0.0, -u2]
    z = np.array([0.0, 0.3, 0.6, 0.9])
    p = np.array([1e-3])
    mu = np.sqrt(1 - z * z)
    I = 1 - u1 * (1 - mu) - u2 * (1 - mu) ** 2
    want = p[0] ** 2 * I / (1 - u1 / 3 - u2 / 6)
    got = clib.lc_deficit(z, p, ld)[:, 0]
    np.testing.assert_allclose(got, want, rtol=2e-5)
    # the star's total flux the oracle normalises by: pi (1 - u1/3 - u2/6); a planet covering the star blocks it all
    assert clib.lc_deficit(np.array([0.0]), np.array([1.0]), ld)[0, 0] == pytest.approx(1.0, abs=1e-12)


def test_oracle_converged_and_hidden_fraction():
    z = np.linspace(0, 1.15, 47)
    p = np.array([0.1, 0.1215])
    a, b = clib.lc_deficit(z, p, LD, nodes=65), clib.lc_deficit(z, p, LD, nodes=257)
    assert np.abs(a - b).max() < 1e-14
    zz, pp = z[:, None], p[None, :]
    np.testing.assert_allclose(clib.lc_hidden(zz, pp), lens_area(zz, pp) / (np.pi * pp ** 2), rtol=0, atol=1e-7)
    assert clib.lc_hidden(0.5, 0.1) == 1.0 and clib.lc_hidden(1.2, 0.1) == 0.0


def test_host_model_against_oracle():
    # the product's numpy model (24-node radial rule) against the oracle's 2-D integral, every z regime
    rp = np.sqrt(0.0146)
    z = np.concatenate([np.linspace(0, 1.2, 61), [1 - rp, 1 + rp, rp, 1.0]])
    p = rp * np.array([0.9, 1.0, 1.1])
    got = 1 - lc.transit_flux(z[:, None], p[None, :], LD)
    np.testing.assert_allclose(got, clib.lc_deficit(z, p, LD), rtol=0, atol=5e-10)
    spec = np.array([0.0144, 0.0146, 0.0149])
    z_tr, hidden = np.array([0.3, 1.05, 11.0, 0.0]), np.array([0.0, 0.0, 1.0, 0.0])
    np.testing.assert_allclose(lc.planet_depths(LD, spec, z_tr, hidden), clib.lc_depths(z_tr, hidden, spec, LD),
                               rtol=0, atol=5e-10)
    np.testing.assert_allclose(lc.uniform_overlap_fraction(z[:, None], p[None, :]),
                               clib.lc_hidden(z[:, None], p[None, :]), rtol=0, atol=1e-7)


def test_uniform_star_matches_lens_formula():
    for p in (0.05, 0.12, 0.3):
        z = np.linspace(0, 1.4, 141)
        num = lc.transit_flux(z, p, [0, 0, 0, 0])
        ana = 1 - lc.uniform_overlap_fraction(z, p) * p * p
        np.testing.assert_allclose(num, ana, rtol=0, atol=2e-9)   # 24-node rule; worst at exact internal contact


def test_limb_darkened_against_brute_force():
    def brute(z, p, n=1500):
        xs = (np.arange(n) + 0.5) / n * 2 * p - p
        X, Y = np.meshgrid(xs, xs)
        r2 = (X + z) ** 2 + Y ** 2
        ok = (X * X + Y * Y <= p * p) & (r2 < 1)
        mu = np.sqrt(np.clip(1 - r2, 0, 1))
        return 1 - (lc.claret_intensity(mu, LD) * ok).sum() * (2 * p / n) ** 2 / lc.stellar_flux_total(LD)
    for z in (0.0, 0.4, 0.9, 1.0, 1.09):
        assert abs(lc.transit_flux(z, 0.12, LD) - brute(z, 0.12)) < 5e-7


def test_limits_and_node_convergence():
    assert lc.transit_flux(1.2, 0.12, LD) == 1.0                       # no contact
    assert lc.transit_flux(0.0, 1.0, LD) == pytest.approx(0.0, abs=1e-9)   # planet covers the star
    z = np.linspace(0, 1.2, 61)
    a = lc.transit_flux(z, 0.1215, LD)
    x, w, d = lc.tanh_sinh_nodes(200, 4.0)
    old = (lc._X, lc._W, lc._D)
    lc._X, lc._W, lc._D = x, w, d
    try:
        b = lc.transit_flux(z, 0.1215, LD)
    finally:
        lc._X, lc._W, lc._D = old
    assert np.abs(a - b).max() < 5e-10                                  # 24 nodes are converged
    assert np.all(np.diff(a[z < 1.12]) >= -1e-12)                        # monotone from centre to limb


def test_orbit_and_transit_timing():
    P, a, inc, T0 = 3.524746, 0.047309 / (1.155 * 0.00465047), 86.71, 2456196.28836
    t = T0 + np.linspace(-0.12, 0.12, 2001)
    z, los = lc.planet_orbit(P, a, 0.0, inc, 0.0, T0, t)
    assert abs(t[np.argmin(z)] - T0) < 2e-4 and los[1000] > 0
    assert abs(z.min() - a * np.cos(np.radians(inc))) < 1e-9             # impact parameter
    f = lc.transit(LD, 0.1209, P, a, 0.0, inc, 0.0, T0, t)
    in_tr = t[f < 1 - 1e-9]
    assert 0.11 < in_tr.max() - in_tr.min() < 0.14                       # T14 of HD 209458 b ~ 3 h
    assert 0.0150 < 1 - f.min() < 0.0175
    # half an orbit later the planet is behind the star: eclipse, no transit
    t2 = t + P / 2
    assert np.all(lc.transit(LD, 0.1209, P, a, 0.0, inc, 0.0, T0, t2) == 1.0)
    e = lc.eclipse(1e-3, 0.1209, P, a, 0.0, inc, 0.0, T0, t2)
    assert e.min() == pytest.approx(1 / 1.001, abs=1e-9) and e.max() == 1.0
    # eccentric orbit: mid-transit still at T0
    z, los = lc.planet_orbit(P, a, 0.3, inc, 40.0, T0, t)
    assert abs(t[np.argmin(np.where(los > 0, z, 99))] - T0) < 5e-4


def test_planet_depths_matrix():
    spec = np.array([0.0144, 0.0146, 0.0149])
    z_tr, hidden = np.array([0.3, 1.05, 11.0]), np.array([0.0, 0.0, 1.0])
    d = lc.planet_depths(LD, spec, z_tr, hidden)
    assert d.shape == (3, 3)
    np.testing.assert_allclose(d[0], 1 - lc.transit_flux(0.3, np.sqrt(spec), LD))
    np.testing.assert_allclose(d[2], spec / (1 + spec))                  # fully eclipsed planet
    assert np.all(d[0] > d[1]) and np.all(np.diff(d[0]) > 0)


def test_jd_to_hjd():
    from wayne_amd import tools
    # the Sun at the March equinox and June solstice of 2000 (Astronomical Almanac low-precision formulae)
    ra, dec = tools.sun_ra_dec(2451623.815)
    assert abs(np.rad2deg(dec)) < 0.02 and min(np.rad2deg(ra), 360 - np.rad2deg(ra)) < 0.02
    ra, dec = tools.sun_ra_dec(2451716.575)
    assert abs(np.rad2deg(ra) - 90) < 0.02 and abs(np.rad2deg(dec) - 23.44) < 0.01
    # a target in the ecliptic swings by +-(1 AU / c) cos(latitude) over the year; at the pole not at all
    jd = np.linspace(2456000.0, 2456365.25, 731)
    au_c = 149597870700.0 / 299792458.0
    d = (tools.jd_to_hjd(jd, 330.795, 18.884) - jd) * 86400.0            # HD 209458: ecliptic latitude 28.7 deg
    assert abs(d.max() - au_c * np.cos(np.deg2rad(28.70))) < 1.0 and abs(d.min() + d.max()) < 1.0
    pole = (tools.jd_to_hjd(jd, 270.0, 66.5607) - jd) * 86400.0            # north ecliptic pole
    assert np.abs(pole).max() < 0.5
    # opposition: the Earth is nearer to the star than the Sun is -> HJD > JD
    k = np.argmax(d)
    ra_s, dec_s = tools.sun_ra_dec(jd[k])
    assert abs(((np.rad2deg(ra_s) - 330.795 + 180) % 360) - 180) > 150
    assert tools.jd_to_hjd(2456196.28836, 330.795, 18.884) > 2456196.28836


def test_observation_uses_heliocentric_times_when_it_knows_the_target():
    from wayne_amd import observation, tools
    args = dict(period=3.524746, sma_au=0.047309, stellar_radius_rsun=1.155, inclination=86.71,
                transittime=2456196.28836)
    t = 2456196.28836 + np.linspace(-0.12, 0.12, 400)
    curves = []
    for coords in (dict(), dict(ra_deg=330.795, dec_deg=18.884)):
        obs = observation.Observation()
        pl = observation.Planet("HD 209458 b", **args, **coords)
        wl = np.linspace(1.0, 1.7, 50)
        obs.setup_target(pl, wl, np.full(50, 0.0146), np.ones(50), ldcoeffs=LD)
        curves.append(obs.generate_lightcurves(t, depth=0.0146)[:, 0])
    shift = float(tools.jd_to_hjd(2456196.28836, 330.795, 18.884) - 2456196.28836)      # days, +403 s here
    # mid-transit in JD moves EARLIER by the correction when the ephemeris is heliocentric
    mid = [np.sum(t * (1 - c)) / np.sum(1 - c) for c in curves]
    assert abs((mid[0] - mid[1]) - shift) < 2e-5 and shift > 0.004
    dd = obs.device_depths(t)
    assert abs(dd.z_tr.argmin() - (1 - curves[1]).argmax()) <= 1


@pytest.mark.gpu
def test_device_depth_matrix_matches_oracle(gpu_ctx):
    import helpers
    from wayne_amd import _lib
    v = helpers.make_visit("small256")
    kw = v.frame_kwargs(0)
    K = len(v.sample_mid_points)
    rng = np.random.RandomState(3)
    z_tr = np.concatenate([np.linspace(0.2, 1.2, K - 2), [0.0, 12.0]])
    hidden = np.where(z_tr > 10, rng.uniform(0, 1, K), 0.0)
    dd = lc.DeviceDepths(z_tr, hidden, v.depth0 * (1 + 0.3 * rng.uniform(-1, 1, v.depth0.size)), LD)
    pg = helpers.product_generator(v, 0)
    from wayne_amd import engine
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    desc = pg.build_descriptor(eng, **dict(kw, planet_signal=dd))
    eng.ctx.upload(0, desc)
    eng.ctx.run_front(0)
    got = eng.ctx.debug_depth(0)
    eng.ctx.run_back(0)
    from wayne_amd import tools
    i0, i1 = tools.crop_spectrum_ind(v.grism.wl_limits[0], v.grism.wl_limits[1], v.wl)
    want = clib.lc_depths(dd.z_tr, dd.hidden, dd.planet_spectrum[i0:i1], LD)     # oracle/lc_oracle.c
    assert got.shape == want.shape
    # float32 integrand on the device: a few 1e-9 of the flux
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-8)
    # and the exposure built from it equals the one built from the uploaded matrix (replay thrower, noise off)
    off = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False)
    a = np.stack([r[0] for r in pg.scanning_frame(rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                  **dict(kw, planet_signal=dd, **off)).reads])
    b = np.stack([r[0] for r in pg.scanning_frame(rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                  **dict(kw, planet_signal=dd.host_matrix(), **off)).reads])
    assert np.abs(a - b).max() < 0.5     # counts may round differently for a handful of bins (depth differs by 1e-9)
    assert np.median(np.abs(a - b)) < 1e-6


@pytest.mark.gpu
def test_device_depths_interpolated_in_radius_ratio(gpu_ctx):
    # a realistic spectrum: radius ratios within 1 % of each other, every z regime (no transit, contacts,
    # planet inside the disk, over the centre): the device interpolates the quadrature in p and must
    # stay within its usual 2e-8 of the numpy model for every wavelength
    import helpers
    from wayne_amd import engine, tools
    v = helpers.make_visit("small256")
    kw = v.frame_kwargs(0)
    K = len(v.sample_mid_points)
    rp0 = np.sqrt(v.depth0.mean())
    z_tr = np.array([1.5, 1.0 + rp0 * 1.0001, 1.0 + rp0 * 0.5, 1.0, 1.0 - rp0 * 0.9999, 0.6, rp0 * 1.0002, rp0 * 0.3, 0.0])
    assert K == z_tr.size
    spec = v.depth0 * (1 + 0.004 * np.sin(np.arange(v.depth0.size) / 300.0))
    dd = lc.DeviceDepths(z_tr, np.zeros(K), spec, LD)
    pg = helpers.product_generator(v, 0)
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    eng.ctx.upload(0, pg.build_descriptor(eng, **dict(kw, planet_signal=dd)))
    eng.ctx.run_front(0)
    got = eng.ctx.debug_depth(0)
    eng.ctx.run_back(0)
    i0, i1 = tools.crop_spectrum_ind(v.grism.wl_limits[0], v.grism.wl_limits[1], v.wl)
    want = clib.lc_depths(dd.z_tr, dd.hidden, dd.planet_spectrum[i0:i1], LD)     # oracle/lc_oracle.c
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-8)
    assert got[0].max() == 0.0 and got[5].min() > 0.01
    # a flat spectrum (one radius ratio for all wavelengths) takes the single-evaluation path
    flat = lc.DeviceDepths(z_tr, np.zeros(K), np.full_like(v.depth0, 0.0146), LD)
    eng.ctx.upload(0, pg.build_descriptor(eng, **dict(kw, planet_signal=flat)))
    eng.ctx.run_front(0)
    got = eng.ctx.debug_depth(0)
    eng.ctx.run_back(0)
    np.testing.assert_allclose(got, clib.lc_depths(flat.z_tr, flat.hidden, flat.planet_spectrum[i0:i1], LD),
                               rtol=0, atol=2e-8)


def test_orbit_over_one_exposure_by_interpolation():
    # depth_inputs evaluates the planet's position at ten Chebyshev points of an exposure's span and interpolates to
    # its thousands of sub-sample times (lightcurve.planet_orbit_short_span): against the direct evaluation, for
    # circular and eccentric orbits, edge-on ones (the separation has a corner at mid-transit), spans up to the
    # routine's limit, regular and irregular sampling, in and out of transit and eclipse
    from wayne_amd import lightcurve as lc
    rng = np.random.default_rng(3)
    worst = 0.0
    used = 0
    for trial in range(200):
        P, a = rng.uniform(0.5, 20), rng.uniform(3, 30)
        e = float(rng.choice([0.0, 0.0, rng.uniform(0, 0.7)]))
        inc = float(rng.choice([90.0, rng.uniform(80, 90)]))
        w, mid = rng.uniform(0, 360), rng.uniform(0, 5)
        K = int(rng.integers(50, 3000))
        span = rng.uniform(1e-5, 0.0039) * P
        t0 = mid + rng.uniform(-1, 1) * P * float(rng.choice([0.0, 0.01, 0.3, 0.5]))
        t = t0 + (np.sort(rng.uniform(0, span, K)) if trial % 2 else np.linspace(0, span, K))
        z1, l1 = lc.planet_orbit(P, a, e, inc, w, mid, t)
        z2, l2 = lc.planet_orbit_short_span(P, a, e, inc, w, mid, t)
        used += int(z2 is not z1)
        worst = max(worst, float(np.abs(z1 - z2).max()), float(np.abs(l1 - l2).max()))
    assert worst < 1e-11, worst           # stellar radii: 1e-12 of a transit depth
    # beyond the limit, or with few samples, the direct evaluation is what runs
    t = 5.0 + np.linspace(0, 0.2, 500)
    np.testing.assert_array_equal(lc.planet_orbit_short_span(3.5, 8.8, 0.1, 87.0, 30.0, 5.05, t)[0],
                                  lc.planet_orbit(3.5, 8.8, 0.1, 87.0, 30.0, 5.05, t)[0])


def test_orbit_interpolation_at_julian_dates_across_exposures():
    # at JD-scale times (ulp 4.7e-10 d) the exposures of a visit share their sub-sample OFFSETS exactly, so
    # the second exposure takes the interpolation matrix cached by the first -- which must have been built around a
    # centre both share (the offsets' own), with node times that are never rounded to the JD grid
    from wayne_amd import lightcurve as lc
    lc._cheb_cache.clear()
    P, a, e, inc, w = 3.52474859, 8.76, 0.0, 86.71, 0.0
    mid = 2455000.0 + 0.813
    off = np.arange(2233) * (124.0 * 2.0 ** -30)              # 10 ms sampling, exactly representable offsets
    worst = []
    for start in (2456001.25, 2456001.25 + 273 * 2.0 ** -12, 2456004.0 + 11 * 2.0 ** -12):   # in transit and out of it
        t = start + off
        assert np.array_equal(t - t[0], off)                  # the JD grid holds these times exactly
        n_before = len(lc._cheb_cache)
        z1, l1 = lc.planet_orbit(P, a, e, inc, w, mid, t)
        z2, l2 = lc.planet_orbit_short_span(P, a, e, inc, w, mid, t)
        assert z2 is not z1
        worst.append(max(float(np.abs(z1 - z2).max()), float(np.abs(l1 - l2).max())))
        if start != 2456001.25:
            assert len(lc._cheb_cache) == n_before            # the first exposure's matrix served this one
    assert max(worst) < 2e-11, worst
    # ... and offsets that the JD grid does NOT hold exactly (a 95-minute cadence): every exposure then differs in the
    # last bits of its offsets, takes a matrix of its own, and is as exact
    for n in range(3):
        t = 2456001.25 + n * (95.0 / 1440.0) + np.arange(2233) * (0.01 / 86400.0)
        z1, l1 = lc.planet_orbit(P, a, e, inc, w, mid, t)
        z2, l2 = lc.planet_orbit_short_span(P, a, e, inc, w, mid, t)
        assert max(float(np.abs(z1 - z2).max()), float(np.abs(l1 - l2).max())) < 2e-11
