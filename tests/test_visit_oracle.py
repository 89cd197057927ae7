"""The visit level against oracle/visit_oracle.py -- an independent statement of the visit
plan, the visit-long trend and the per-exposure bookkeeping of Observation._generate_exposure (observation.py
:197-291, 415-504; visit_planner.py; visit_trends.py; tools.py:274-300) with oracle/lc_oracle.c as its light-curve
model -- first pinned to the reference's own test values, then compared with the product: host logic on the CPU,
whole exposures on the GPU."""
import numpy as np
import pytest

import helpers
from oracle import visit_oracle as vo, wayne_oracle as wo
from wayne_amd import _lib, detector, lightcurve, observation, visit_planner
from wayne_amd.trend_generators import scan_speed_varations, visit_trends

LD = [0.800627, -0.757066, 0.897268, -0.384804]
ORBIT = dict(period=3.524746, sma_au=0.047309, stellar_radius_rsun=1.155, inclination=86.71, transittime=2456196.28836)


def test_oracle_pinned_to_the_reference_test_values():
    # tests/trend_generators/test_visit_trends.py:38-52
    t = np.array([6, 9, 12, 95, 98, 101]) / 1440.0
    got = vo.hook_and_long_term_ramp(t, [0, 3], 0.005, 0.0011, 400, 9 / 60 / 24)
    np.testing.assert_array_almost_equal(got, [0.99891, 0.99952, 0.99978, 0.9986, 0.99921, 0.99947], decimal=5)
    # tests/test_tools.py:85-89
    assert vo.detect_orbits([1.001, 1.002, 1.032]) == [0, 2]
    # detector.py:269-297 (tests/test_detector.py has no value for it: hand-computed 2*16*4 // 6, and the 304 cap)
    assert vo.num_exp_per_buffer(5, 256) == 21 and vo.num_exp_per_buffer(2, 64) == 101


@pytest.mark.parametrize("mode", [(5, "SPARS10", 256, 3), (16, "SPARS10", 1024, 4), (4, "RAPID", 64, 2)])
def test_visit_plan_and_trend_against_the_oracle(mode):
    NSAMP, SAMPSEQ, SUBARRAY, n_orb = mode
    det = detector.WFC3_IR()
    want = vo.visit_planner(det, NSAMP, SAMPSEQ, SUBARRAY, n_orb, exp_overhead=3.)
    got = visit_planner.VisitPlanner(det, NSAMP, SAMPSEQ, SUBARRAY, num_orbits=n_orb, exp_overhead=3.)
    np.testing.assert_allclose(got["exp_times"], want["exp_times"], rtol=0, atol=1e-12)
    assert list(got["orbit_start_index"]) == want["orbit_start_index"] and got["num_exp"] == want["num_exp"]
    assert list(got["buffer_dump_index"]) == want["buffer_dump_index"]
    starts = want["exp_times"] / 1440.0 + 2456196.1
    plan = {"exp_start_times": starts, "orbit_start_index": want["orbit_start_index"]}
    coeffs = (0.005, 0.0011, 400, starts[2])
    np.testing.assert_allclose(visit_trends.HookAndLongTermRamp(plan, coeffs).scale_factors,
                               vo.hook_and_long_term_ramp(starts, want["orbit_start_index"], *coeffs), rtol=1e-14)


@pytest.mark.parametrize("e,w", [(0.0, 0.0), (0.3, 40.0), (0.6, 200.0)])
def test_orbit_against_the_oracle(e, w):
    P, a, inc, T0 = 3.524746, 8.81, 86.71, 2456196.28836
    t = T0 + np.linspace(-2.0, 2.0, 4001)
    z, los = lightcurve.planet_orbit(P, a, e, inc, w, T0, t)
    zo, front = vo.separation(P, a, e, inc, w, T0, t)
    np.testing.assert_allclose(z, zo, rtol=0, atol=2e-8)      # (JD ~ 2.4e6 d: one ulp of the time is 5e-10 d)
    assert np.array_equal(los > 0, front) or np.abs(los[(los > 0) != front]).max() < 1e-9
    i = np.argmin(np.where(front, zo, 99.0))
    assert abs(t[i] - T0) < 2e-3                               # mid-transit where it should be


def make_pair(name, n_exp, spatial_scan, **obs_kw):
    """A product Observation and its oracle twin over the same synthetic inputs."""
    v = helpers.make_visit(name, n_exposures=n_exp)
    det = v.detector
    rng = np.random.RandomState(4)
    x_ref = v.cfg["x_ref"] + rng.uniform(-0.5, 0.5, n_exp)    # per-exposure arrays, as the example's xref.txt / sky.txt
    sky = rng.uniform(0.5, 2.0, n_exp)
    y_ref = v.cfg["y_ref"] - (20.0 if spatial_scan else 0.0)
    planet = observation.Planet("oracle-test", rp_over_rs=0.1209, **ORBIT)
    spectrum = v.depth0
    obs = observation.Observation(calibration=v.calibration, seed=v.seed)
    obs.setup_detector(det, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    obs.setup_grism(v.grism)
    obs.setup_target(planet, v.wl, spectrum, v.stellar_flux, ldcoeffs=LD)
    obs.setup_visit(2456196.25, 2)
    obs.setup_observation(x_ref, y_ref, spatial_scan=spatial_scan, scan_speed=v.scan_speed)
    obs.setup_simulator(sample_rate=600.0)
    obs.setup_reductions()
    ssv = scan_speed_varations.SSVSine(1.5, 1.1, 0.0) if spatial_scan else None
    obs.setup_trends(ssv, x_shifts=0.013, y_shifts=-0.004, x_jitter=0.02, y_jitter=0.01)
    obs.setup_noise_sources(sky_background=sky, cosmic_rate=None, add_read_noise=False, add_stellar_noise=False)
    coeffs = (0.005, 0.0011, 400, obs.exp_start_times[1])
    obs.setup_visit_trend(coeffs)
    obs.transmission_spectroscopy = True
    _, _, eo = wo.from_calibration(v.calibration, v.grism.name, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    orbit = (ORBIT["period"], planet.sma_over_rs, 0.0, ORBIT["inclination"], 0.0, ORBIT["transittime"])
    oo = vo.ObservationOracle(eo, det, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, v.wl, v.stellar_flux, spectrum, orbit, LD,
                              0.1209, x_ref, y_ref, spatial_scan, v.scan_speed, 600.0, 2456196.25, 2,
                              x_shifts=0.013, y_shifts=-0.004, x_jitter=0.02, y_jitter=0.01, sky_background=sky,
                              visit_trend_coeffs=coeffs,
                              frame_kwargs=dict(ssv_generator=wo.SSVSine(1.5, 1.1, 0.0) if spatial_scan else None,
                                                cosmic_rate=None, add_read_noise=False, add_stellar_noise=False,
                                                add_dark=True, add_flat=True))
    return v, obs, oo


def test_exposure_bookkeeping_against_the_oracle():
    v, obs, oo = make_pair("tiny", 6, True)
    np.testing.assert_allclose(obs.exp_start_times, oo.exp_start_times, rtol=0, atol=1e-12)
    np.testing.assert_allclose(obs._visit_trend.scale_factors, oo.scale_factors, rtol=1e-14)
    assert len(oo.exp_start_times) >= 6
    for number in (1, 2, 5):
        inp = oo.exposure_inputs(number)
        i = number - 1
        assert inp["x_ref"] == obs._try_index(obs.x_ref, i) + obs.x_shifts * i
        assert inp["y_ref"] == obs.y_ref + obs.y_shifts * i and inp["sky_background"] == obs.sky_background[i]
        # the depths the device is asked to compute, evaluated by the product's host model, vs the oracle's
        dd = obs.device_depths(inp["time_array"])
        np.testing.assert_allclose(dd.host_matrix(), inp["planet_signal"], rtol=0, atol=5e-10)
    assert oo.exposure_inputs(2)["planet_signal"].max() > 0.01          # the visit covers the transit


@pytest.mark.gpu
@pytest.mark.parametrize("name,scan", [("tiny", True), ("stare256", False)])
def test_generated_exposures_against_the_oracle(name, scan):
    # Observation._generate_exposure -> reads, for exposures in the middle of a visit (shifted x / y, own sky, own
    # point of the visit ramp, light curve evaluated on the device), against the oracle's statement of the same
    # exposure: bit-exact replay thrower, float64 reads, deterministic stages + sky + dark
    v, obs, oo = make_pair(name, 5, scan)
    obs.frame_options = dict(rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64, exact_samplers=True, threads=2)
    N = v.detector.light_sensitive_size(v.SUBARRAY)
    for number in (2, 4):
        exp = obs._generate_exposure(obs.exp_start_times[number - 1], number, write_fits=False)
        got = np.stack([r[0] for r in exp.reads])
        want = np.stack(oo.generate_exposure(number, wo.PhiloxDraws(v.seed, number - 1, N), threads=2, thrower="oracle"))
        assert got.shape == want.shape
        d = np.abs(got - want)
        # (the device's light curve is good to 2e-8: a bin's np.round may land on the other side for a few bins)
        bad = int((d > 1e-3 + 1e-6 * np.abs(want)).sum())
        assert bad <= 2e-3 * got.size, "exposure %d: %d of %d pixels differ" % (number, bad, got.size)
        assert np.median(d) < 1e-4 and got[-1].max() > 10
    from wayne_amd import engine
    engine.close_all()
