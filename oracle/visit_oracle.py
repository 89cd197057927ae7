"""oracle/visit_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of what sits ABOVE ExposureGenerator on the path: the visit plan, the visit-long trend and
the per-exposure bookkeeping of Observation._generate_exposure (reference: wayne/observation.py:197-291,
388-514; wayne/visit_planner.py:5-129; wayne/trend_generators/visit_trends.py:10-73; wayne/tools.py:274-300).
Units are plain floats: times in days (JD) unless a name says otherwise, the planner's internal clock in
minutes as in the reference.

Pinned by the reference's own test values for the ramp (tests/trend_generators/test_visit_trends.py:38-52) and
for detect_orbits (tests/test_tools.py:85-89) in tests/test_reference_goldens.py / tests/test_visit_oracle.py;
the rest is PARITY UNPINNED (the reference has no test above the exposure level) and restated line against line.

The orbit -> projected separation step is pylightcurve's (an absent third-party dependency, setup.py:33),
restated from its published convention: true anomaly f from Kepler's equation (solved here by bisection, on
purpose not the product's Newton iteration), mid-transit at f = pi/2 - omega,
    r = a (1 - e^2) / (1 + e cos f),   z = r sqrt(1 - sin^2(omega + f) sin^2 i),
the planet in front of the star while sin(omega + f) > 0.
"""
import numpy as np

from . import clib
from . import wayne_oracle as wo


def visit_planner(detector, NSAMP, SAMPSEQ, SUBARRAY, num_orbits=3, time_per_orbit=54., hst_period=95.,
                  exp_overhead=1.):
    """visit_planner.py:5-129 (minutes).  `detector` needs exptime() [s] and num_exp_per_buffer()."""
    exptime_min = detector.exptime(NSAMP, SUBARRAY, SAMPSEQ) / 60.
    exp_per_dump = num_exp_per_buffer(NSAMP, SUBARRAY)
    time_buffer_dump = 5.8                                   # :78
    exp_times, orbit_start_index, buffer_dump_index = [], [], []
    for orbit_n in range(num_orbits):                        # :89
        guide_star_aq = 6. if orbit_n == 0 else 5.           # :90-93
        orbit_start_index.append(len(exp_times))             # :96
        start_time = hst_period * orbit_n
        visit_time = start_time + guide_star_aq
        visit_end_time = start_time + time_per_orbit
        exp_n = 0
        while visit_time < visit_end_time:                   # :104
            exp_times.append(visit_time)
            visit_time += exptime_min + exp_overhead
            exp_n += 1
            if exp_n > exp_per_dump:                         # :113-117
                visit_time += time_buffer_dump
                exp_n = 0
                buffer_dump_index.append(len(exp_times))
    return {"exp_times": np.array(exp_times), "num_exp": len(exp_times), "orbit_start_index": orbit_start_index,
            "buffer_dump_index": buffer_dump_index, "exptime": exptime_min * 60.}


def num_exp_per_buffer(NSAMP, SUBARRAY):
    """detector.py:269-297: the buffer holds 2 full-frame 16-read exposures, at most 304 reads' headers."""
    hard_limit = 304
    headers_per_exp = NSAMP + 1
    total_allowed_reads = 2 * 16 * (1024 // SUBARRAY)
    if total_allowed_reads > hard_limit:
        total_allowed_reads = hard_limit
    return int(np.floor(total_allowed_reads / headers_per_exp))


def detect_orbits(exp_start_times, separation=0.028):
    """tools.py:274-300."""
    exp_start_times = np.array(exp_start_times)
    last = exp_start_times[0]
    orbit_index = [0]
    for i, t in enumerate(exp_start_times):
        if t - last >= separation:
            orbit_index.append(i)
        last = t
    return orbit_index


def orbit_start_times_per_exp(time_array, obs_start_index):
    """visit_trends.py:60-73."""
    obs_index = list(obs_start_index) + [len(time_array)]
    t_0 = np.zeros(len(time_array))
    for i in range(len(obs_index) - 1):
        t_0[obs_index[i]:obs_index[i + 1]] = time_array[obs_start_index[i]]
    return t_0


def hook_and_long_term_ramp(exp_start_times, orbit_start_index, a1, b1, b2, to):
    """visit_trends.py:35-57: (1 - a1 (t - to)) (1 - b1 exp(-b2 (t - t_0)))."""
    t = np.array(exp_start_times, dtype=float)
    t_0 = orbit_start_times_per_exp(t, orbit_start_index)
    return (1 - a1 * (t - to)) * (1 - b1 * np.exp(-b2 * (t - t_0)))


def true_anomaly(mean_anomaly, e):
    """Kepler's equation E - e sin E = M by bisection on [M - e, M + e] (60 halvings), then f from E."""
    M = np.asarray(mean_anomaly, dtype=float)
    lo, hi = M - e - 1e-12, M + e + 1e-12
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        too_small = mid - e * np.sin(mid) < M
        lo = np.where(too_small, mid, lo)
        hi = np.where(too_small, hi, mid)
    E = 0.5 * (lo + hi)
    return 2.0 * np.arctan2(np.sqrt(1 + e) * np.sin(E / 2), np.sqrt(1 - e) * np.cos(E / 2))


def separation(period, sma_over_rs, e, inclination_deg, periastron_deg, mid_time, times):
    """(z in stellar radii, in_front) at each time."""
    inc, w = np.radians(inclination_deg), np.radians(periastron_deg)
    f_tr = 0.5 * np.pi - w
    E_tr = 2.0 * np.arctan(np.sqrt((1 - e) / (1 + e)) * np.tan(0.5 * f_tr))
    M_tr = E_tr - e * np.sin(E_tr)
    M = M_tr + 2 * np.pi * (np.asarray(times, dtype=float) - mid_time) / period
    M = (M + np.pi) % (2 * np.pi) - np.pi
    f = true_anomaly(M, e)
    r = sma_over_rs * (1 - e * e) / (1 + e * np.cos(f))
    s = np.sin(w + f)
    return r * np.sqrt(1 - s * s * np.sin(inc) ** 2), s > 0


def planet_depths(orbit, ldcoeffs, planet_spectrum, times, rp_white):
    """observation.py:293-357, 442-443: 1 - [transit(Rp/Rs = sqrt(depth)) - (1 - eclipse(depth, rp))] per
    sub-sample and wavelength, with oracle/lc_oracle.c as the light-curve model."""
    z, front = separation(*orbit, times)
    z_tr = np.where(front, z, 10.0 + z)                      # behind the star: no transit
    hidden = np.where(front, 0.0, clib.lc_hidden(z, np.full(z.shape, rp_white)))
    return clib.lc_depths(z_tr, hidden, planet_spectrum, ldcoeffs)


def try_index(value, index):
    """observation.py:506-514."""
    try:
        return value[index]
    except TypeError:
        return value


class ObservationOracle(object):
    """The attributes Observation.setup_* collect (observation.py:46-291), as keyword arguments, and the two
    things run_observation does with them: the start times (setup_visit) and one exposure (_generate_exposure)."""

    def __init__(self, exposure_oracle, detector, NSAMP, SAMPSEQ, SUBARRAY, wl, stellar_flux, planet_spectrum,
                 orbit, ldcoeffs, rp_white, x_ref, y_ref, spatial_scan, scan_speed, sample_rate, start_JD, num_orbits,
                 exp_start_times=None, x_shifts=0., y_shifts=0., x_jitter=1e-7, y_jitter=1e-7, sky_background=1.0,
                 visit_trend_coeffs=None, frame_kwargs=None):
        self.eo, self.detector = exposure_oracle, detector
        self.NSAMP, self.SAMPSEQ, self.SUBARRAY = NSAMP, SAMPSEQ, SUBARRAY
        self.wl, self.stellar_flux, self.planet_spectrum = wl, stellar_flux, planet_spectrum
        self.orbit, self.ldcoeffs, self.rp_white = orbit, ldcoeffs, rp_white
        self.x_ref, self.y_ref, self.sky_background = x_ref, y_ref, sky_background
        self.spatial_scan, self.scan_speed, self.sample_rate = spatial_scan, scan_speed, sample_rate
        self.x_shifts, self.y_shifts, self.x_jitter, self.y_jitter = x_shifts, y_shifts, x_jitter, y_jitter
        self.frame_kwargs = frame_kwargs or {}
        if exp_start_times is not None:                       # observation.py:202-204, 227-237
            self.exp_start_times = np.asarray(exp_start_times, dtype=float)
            self.orbit_start_index = detect_orbits(self.exp_start_times)
        else:                                                 # :209-225 (exp_overhead = 3 min "to make observations sparser")
            plan = visit_planner(detector, NSAMP, SAMPSEQ, SUBARRAY, num_orbits, exp_overhead=3.)
            self.exp_start_times = plan["exp_times"] / (60. * 24.) + start_JD
            self.orbit_start_index = plan["orbit_start_index"]
        self.scale_factors = None
        if visit_trend_coeffs is not None:                    # :279-291
            self.scale_factors = hook_and_long_term_ramp(self.exp_start_times, self.orbit_start_index,
                                                         *visit_trend_coeffs)

    def exposure_inputs(self, number, with_depths=True):
        """What _generate_exposure hands to scanning_frame / staring_frame for file number `number`
        (observation.py:415-462).  `with_depths=False` leaves the K x W light-curve matrix out (None): this oracle's
        model is a 2-D integration per sample and wavelength, minutes for an in-transit exposure of 2233 sub-samples."""
        index_number = number - 1
        expstart = self.exp_start_times[index_number]
        sample_rate = self.sample_rate if self.spatial_scan else 365.25 * 86400. * 1000.   # :433-434 (1 yr, in ms)
        _, mids, durs, read_index = self.eo._gen_scanning_sample_times(sample_rate)          # :436
        time_array = expstart + mids / (86400. * 1000.)                                        # :439
        depths = planet_depths(self.orbit, self.ldcoeffs, self.planet_spectrum, time_array, self.rp_white) if with_depths else None
        x_ref = try_index(self.x_ref, index_number) + self.x_shifts * index_number            # :449-455
        y_ref = try_index(self.y_ref, index_number) + self.y_shifts * index_number
        sky = try_index(self.sky_background, index_number)
        scale = None if self.scale_factors is None else self.scale_factors[index_number]       # :459-462
        return dict(x_ref=x_ref, y_ref=y_ref, sky_background=sky, scale_factor=scale, planet_signal=depths,
                    sample_mid_points=mids, sample_durations=durs, read_index=read_index, time_array=time_array)

    def generate_exposure(self, number, draws, **oracle_kw):
        """The reads of exposure `number` (observation.py:464-500) through ExposureOracle."""
        # `planet_signal`: a K x W depth matrix to use instead of this oracle's own light curves (a test hands in the
        # device's, after comparing the two, so that np.round / Poisson of the counts see the same means to the last bit)
        override = oracle_kw.pop("planet_signal", None)
        inp = self.exposure_inputs(number, with_depths=override is None)
        kw = dict(self.frame_kwargs)
        kw.update(sky_background=inp["sky_background"], scale_factor=inp["scale_factor"])
        kw.update(oracle_kw)
        if override is not None:
            assert np.shape(override) == (len(inp["sample_mid_points"]), len(self.wl))
            inp["planet_signal"] = np.asarray(override, dtype=float)
        if self.spatial_scan:
            return self.eo.scanning_frame(inp["x_ref"], inp["y_ref"], self.x_jitter, self.y_jitter, self.wl,
                                          self.stellar_flux, inp["planet_signal"], self.scan_speed, self.sample_rate,
                                          inp["sample_mid_points"], inp["sample_durations"], inp["read_index"],
                                          draws=draws, **kw)
        kw.pop("ssv_generator", None)                          # staring_frame takes none (exposure_generator.py:146-153)
        return self.eo.staring_frame(inp["x_ref"], inp["y_ref"], self.x_jitter, self.y_jitter, self.wl,
                                     self.stellar_flux, inp["planet_signal"], inp["sample_mid_points"],
                                     inp["sample_durations"], inp["read_index"], draws=draws, **kw)


# ---------------------------------------------------------------------------
# the CLI's ingestion of a parameter file (run_visit.py:41-320), for the reference's example visit
# ---------------------------------------------------------------------------
R_SUN_IN_AU = 6.957e8 / 1.495978707e11      # IAU 2015 nominal solar radius over the astronomical unit


def blackbody_lambda(wl_um, T):
    """Planck's law per unit wavelength in erg / (s cm^2 angstrom sr), what astropy's blackbody_lambda returns
    (run_visit.py:201-203); wavelengths in micron.  Written from the SI form, converted at the end."""
    h, c, kB = 6.62607015e-34, 2.99792458e8, 1.380649e-23
    lam = np.asarray(wl_um, dtype=float) * 1e-6                                  # m
    b_si = 2.0 * h * c ** 2 / lam ** 5 / (np.exp(h * c / (lam * kB * T)) - 1.0)  # W m^-2 m^-1 sr^-1
    return b_si * 1e7 * 1e-4 * 1e-10                                             # erg/s, per cm^2, per angstrom


def visit_from_parameter_file(cfg, directory, exposure_oracle, detector, star_temperature=6100.0):
    """The ObservationOracle of a parsed YAML parameter file whose file names are relative to `directory`, following
    run_visit.py statement by statement for the keys the example file holds: planet spectrum sorted by wavelength
    and cropped to 0.9-1.8 micron whatever the grism (:151-153), no rebinning (`rebin_resolution: false`), stellar
    flux = black body x flux_scale when there is no stellar spectrum file to read (:201-205; the example's FITS blob
    is absent here and its YAML says the flux is a black body), per-exposure x_ref / y_ref / sky / start times from
    text files (:268-290), SSVSine from `ssv_coeffs` (:233-245), the visit ramp (:316-318).  The orbit's a / R* from
    `sma` (au) and `stellar_radius` (solar radii); Rp/R* of the white light curve = sqrt(mean depth) (the reference
    takes it from the Open Exoplanet Catalogue entry, which is absent: it only enters the secondary eclipse).
    Returns (ObservationOracle, dict of the plain inputs)."""
    import os
    target, oc = cfg["target"], cfg["observation"]

    def load(name):
        return np.loadtxt(os.path.join(directory, name))

    spec = load(target["planet_spectrum_file"])
    order = np.argsort(spec[:, 0], kind="stable")                       # tools.load_and_sort_spectrum (tools.py:182-200)
    wl_all, depth_all = spec[order, 0], spec[order, 1]
    i0, i1 = wo.crop_spectrum_ind(0.9, 1.8, wl_all.copy())              # run_visit.py:152-153
    wl, depth = wl_all[i0:i1], depth_all[i0:i1]
    assert not target["rebin_resolution"]
    flux = blackbody_lambda(wl, star_temperature) * target["flux_scale"]
    sma_over_rs = target["sma"] / (target["stellar_radius"] * R_SUN_IN_AU)
    orbit = (target["period"], sma_over_rs, target["eccentricity"], target["inclination"], target["periastron"],
             target["transit_time"])
    x_ref = load(oc["x_ref"]) if isinstance(oc["x_ref"], str) else oc["x_ref"]
    y_ref = load(oc["y_ref"]) if isinstance(oc["y_ref"], str) else oc["y_ref"]
    sky = load(oc["sky_background"]) if isinstance(oc["sky_background"], str) else oc["sky_background"]
    starts = load(oc["exp_start_times"]) if oc.get("exp_start_times") else None
    assert oc["ssv_type"] == "sine" and oc["spatial_scan"]
    frame_kwargs = dict(ssv_generator=wo.SSVSine(*oc["ssv_coeffs"]), cosmic_rate=oc["cosmic_rate"],
                        noise_mean=oc["noise_mean"], noise_std=oc["noise_std"], add_dark=oc["add_dark"],
                        add_flat=oc["add_flat"], add_gain_variations=oc["add_gain_variations"],
                        add_non_linear=oc["add_non_linear"], add_read_noise=oc["add_read_noise"],
                        add_initial_bias=oc["add_initial_bias"], add_stellar_noise=oc["add_stellar_noise"],
                        clip_values_det_limits=oc["clip_values_det_limits"], threads=cfg["general"]["threads"])
    rp_white = float(np.sqrt(np.mean(depth)))
    oo = ObservationOracle(exposure_oracle, detector, oc["NSAMP"], oc["SAMPSEQ"], oc["SUBARRAY"], wl, flux, depth, orbit,
                           target["ldcoeffs"], rp_white, x_ref, y_ref, True, oc["scan_speed"], oc["sample_rate"],
                           oc["start_JD"] or 0.0, oc["num_orbits"], exp_start_times=starts, x_shifts=oc["x_shifts"],
                           y_shifts=oc["y_shifts"], x_jitter=oc["x_jitter"], y_jitter=oc["y_jitter"], sky_background=sky,
                           visit_trend_coeffs=cfg["trends"]["visit_trend_coeffs"], frame_kwargs=frame_kwargs)
    return oo, dict(wl=wl, depth=depth, flux=flux, x_ref=x_ref, y_ref=y_ref, sky=sky, starts=starts, orbit=orbit,
                    rp_white=rp_white)
