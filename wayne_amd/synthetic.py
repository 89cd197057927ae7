"""Synthetic visits for benchmarks, smoke and parity tests.

Nothing here is fetched: the stellar spectrum FITS of the reference's example
is not in its tree and pylightcurve / pysynphot are not installed, so the
inputs are generated from seeds following SURVEY.md section 8(d):

  wavelength grid   wl = sort(1 / (0.5 + 1e-4 arange(15000))) micron -- the grid of
                    examples/hd209458b_12181_simulation_planetary_spectrum.dat
                    (constant step in 1/lambda, 0.5-2.0 micron)
  stellar flux      6100 K black-body shape, scaled so an exposure holds E electrons
  transit depth     0.0146 + 2e-4 sin(2 pi (wl - 1.1) / 0.3), times a smooth trapezoid in time
  geometry / modes  per config, below (BASELINE.json `configs`)

CONFIGS
  cfg2  G141 staring,      SUBARRAY 1024, SPARS10, NSAMP 16, K = 15,  E = 2.5e7
  cfg3  G141 spatial scan, SUBARRAY 256,  SPARS10, NSAMP 15, K = 64,  E = 4e8,  1.6 px/s
  cfg4  G141 spatial scan, SUBARRAY 1024, SPARS10, NSAMP 16, K = 128, E = 1e9,  5 px/s   <- the metric's config
  cfg5  cfg4 for G141 then G102, SSVSine(1.5, 1.1, 0), cosmic rate 11, sky 4.7-6.7
  tiny  G141 scan, SUBARRAY 64, RAPID, NSAMP 4, K = 6, E = 3e4   (parity / smoke)
"""
import numpy as np

from . import tools

H_C_OVER_K = 1.438776877e4   # micron * K


def wavelength_grid():
    return np.sort(1.0 / (0.5 + 1e-4 * np.arange(15000)))


def blackbody_shape(wl_um, T=6100.0):
    """B_lambda up to a constant (per unit wavelength)."""
    wl = np.asarray(wl_um, dtype=float)
    return 1.0 / (wl ** 5 * (np.exp(H_C_OVER_K / (wl * T)) - 1.0))


def depth_spectrum(wl_um):
    wl = np.asarray(wl_um, dtype=float)
    return 0.0146 + 2e-4 * np.sin(2 * np.pi * (wl - 1.1) / 0.3)


def transit_shape(t_days, mid=0.0, full_width=0.125, ingress=0.02):
    """Smooth trapezoid g(t) in [0, 1]: 1 in transit, cosine ingress / egress."""
    t = np.abs(np.asarray(t_days, dtype=float) - mid)
    half = full_width / 2.
    g = np.where(t <= half - ingress, 1.0, 0.0)
    edge = (t > half - ingress) & (t < half)
    g = np.where(edge, 0.5 * (1 + np.cos(np.pi * (t - (half - ingress)) / ingress)), g)
    return g


def sample_times(read_times_s, K_per_read=None, sample_rate_ms=None):
    """Sub-sample (mid points, durations, read_index) in ms: either the
    reference's fixed-rate sampling (exposure_generator.py:531-579) or exactly
    K_per_read[r] equal sub-samples inside read interval r."""
    read_ms = np.asarray(read_times_s, dtype=float) * 1000.
    if sample_rate_ms is not None:
        starts, idx, i, prev = [], [], -1, 0.
        for rt in read_ms:
            s = np.arange(prev, rt, sample_rate_ms)
            starts.append(s)
            i += len(s)
            idx.append(i)
            prev = rt
        starts = np.concatenate(starts)
        ends = np.roll(starts, -1)
        ends[-1] = read_ms[-1]
    else:
        starts, ends, idx, prev, i = [], [], [], 0., -1
        for rt, n in zip(read_ms, K_per_read):
            e = np.linspace(prev, rt, n + 1)
            starts.append(e[:-1])
            ends.append(e[1:])
            i += n
            idx.append(i)
            prev = rt
        starts, ends = np.concatenate(starts), np.concatenate(ends)
    durs = ends - starts
    return starts + durs / 2., durs, idx


def distribute_samples(read_times_s, K):
    """K sub-samples over the reads, in proportion to each read interval, at least one each."""
    dt = np.diff(np.concatenate([[0.], np.asarray(read_times_s, dtype=float)]))
    R = len(dt)
    if K < R:
        raise ValueError("need at least one sub-sample per read")
    n = np.ones(R, dtype=int)
    rest = K - R
    share = dt / dt.sum() * rest
    n += np.floor(share).astype(int)
    for j in np.argsort(-(share - np.floor(share)))[:K - n.sum()]:
        n[j] += 1
    return n


CONFIGS = {
    # BASELINE.json configs[0]: the shape of the reference's example visit (examples/...parameters.yml:30-47):
    # 10 ms sampling of a 22.3 s exposure -> K = 2233 sub-samples
    "cfg1": dict(grism="G141", SUBARRAY=256, SAMPSEQ="SPARS10", NSAMP=5, E=2.5e7, scan_speed=7.4325,
                 x_ref=404.0, y_ref=457.3 - 70.0, sample_rate=10.0, ssv=(1.5, 1.1, 0.0), cosmic_rate=11.0),
    "cfg2": dict(grism="G141", SUBARRAY=1024, SAMPSEQ="SPARS10", NSAMP=16, K=15, E=2.5e7, scan_speed=0.0,
                 x_ref=404.5 + 379, y_ref=500.0),
    "cfg3": dict(grism="G141", SUBARRAY=256, SAMPSEQ="SPARS10", NSAMP=15, K=64, E=4e8, scan_speed=1.6,
                 x_ref=404.5, y_ref=457.4 - 75.0),
    "cfg4": dict(grism="G141", SUBARRAY=1024, SAMPSEQ="SPARS10", NSAMP=16, K=128, E=1e9, scan_speed=5.0,
                 x_ref=404.5 + 379, y_ref=80.0),
    "cfg5": dict(grism="G141", SUBARRAY=1024, SAMPSEQ="SPARS10", NSAMP=16, K=128, E=1e9, scan_speed=5.0,
                 x_ref=404.5 + 379, y_ref=80.0, ssv=(1.5, 1.1, 0.0), cosmic_rate=11.0),
    "cfg5_g102": dict(grism="G102", SUBARRAY=1024, SAMPSEQ="SPARS10", NSAMP=16, K=128, E=1e9, scan_speed=5.0,
                      x_ref=404.5 + 379, y_ref=80.0, ssv=(1.5, 1.1, 0.0), cosmic_rate=11.0),
    "tiny": dict(grism="G141", SUBARRAY=64, SAMPSEQ="RAPID", NSAMP=4, K=6, E=3e4, scan_speed=40.0,
                 x_ref=440.0, y_ref=490.0, n_wl=600),
    "tiny_g102": dict(grism="G102", SUBARRAY=64, SAMPSEQ="RAPID", NSAMP=3, K=4, E=2e4, scan_speed=25.0,
                      x_ref=455.0, y_ref=492.0, n_wl=500),
    "tiny128": dict(grism="G141", SUBARRAY=128, SAMPSEQ="RAPID", NSAMP=5, K=8, E=1e5, scan_speed=60.0,
                    x_ref=400.0, y_ref=460.0, n_wl=900),
    "tiny512": dict(grism="G141", SUBARRAY=512, SAMPSEQ="RAPID", NSAMP=3, K=5, E=3e5, scan_speed=8.0,
                    x_ref=330.0, y_ref=420.0, n_wl=1500),
    "stare256": dict(grism="G141", SUBARRAY=256, SAMPSEQ="SPARS10", NSAMP=4, K=3, E=1e6, scan_speed=0.0,
                     x_ref=404.5, y_ref=500.0),
    "small256": dict(grism="G141", SUBARRAY=256, SAMPSEQ="SPARS10", NSAMP=4, K=9, E=3e6, scan_speed=3.0,
                     x_ref=404.5, y_ref=420.0),
}


class Visit(object):
    """Inputs of a synthetic visit: everything scanning_frame takes, per exposure."""

    def __init__(self, name, detector, grism, calibration, n_exposures=1, seed=1963, E=None, K=None):
        cfg = dict(CONFIGS[name])
        self.name, self.cfg = name, cfg
        self.detector, self.grism, self.calibration = detector, grism, calibration
        self.seed = seed
        self.n_exposures = n_exposures
        self.NSAMP, self.SAMPSEQ, self.SUBARRAY = cfg["NSAMP"], cfg["SAMPSEQ"], cfg["SUBARRAY"]
        self.E = E or cfg["E"]
        self.scan_speed = cfg["scan_speed"]
        self.read_times = detector.get_read_times(self.NSAMP, self.SUBARRAY, self.SAMPSEQ)
        if "sample_rate" in cfg:    # the reference's fixed-rate sampling
            self.sample_mid_points, self.sample_durations, self.read_index = sample_times(
                self.read_times, sample_rate_ms=cfg["sample_rate"])
            self.K = len(self.sample_mid_points)
        else:
            self.K = K or cfg["K"]
        wl = wavelength_grid()
        if "n_wl" in cfg:   # thinner grid for tiny cases
            i0, i1 = tools.crop_spectrum_ind(grism.wl_limits[0], grism.wl_limits[1], wl)
            sel = np.linspace(i0, i1 - 1, cfg["n_wl"]).astype(int)
            wl = wl[np.unique(sel)]
        self.wl = wl
        if "sample_rate" not in cfg:
            n_per_read = distribute_samples(self.read_times, self.K)
            self.sample_mid_points, self.sample_durations, self.read_index = sample_times(self.read_times, n_per_read)
        # scale the black body so that one exposure throws E electrons
        i0, i1 = tools.crop_spectrum_ind(grism.wl_limits[0], grism.wl_limits[1], wl)
        cw = wl[i0:i1]
        sens_wl, sens_val = calibration.sensitivity(grism.name)
        per_flux_unit = (blackbody_shape(cw) * np.interp(cw, sens_wl, sens_val) * tools.bin_centers_to_widths(cw) * 1e4
                         ).sum() * (self.read_times[-1])
        self.stellar_flux = blackbody_shape(wl) * (self.E / per_flux_unit)
        self.depth0 = depth_spectrum(wl)
        # exposures spread over +-0.1 d around mid-transit
        self.exp_start_days = np.linspace(-0.1, 0.1, n_exposures) if n_exposures > 1 else np.array([0.0])
        rng = np.random.RandomState(seed)
        self.x_refs = cfg["x_ref"] + rng.uniform(-0.5, 0.5, n_exposures)
        self.y_refs = cfg["y_ref"] + rng.uniform(-0.1, 0.1, n_exposures)
        self.sky = rng.uniform(4.7, 6.7, n_exposures)           # examples/...sky.txt range
        self.x_jitter, self.y_jitter = 0.025, 1e-15              # yml:45-47
        self.cosmic_rate = cfg.get("cosmic_rate", 11.0)
        self.ssv = cfg.get("ssv")

    def planet_signal(self, i):
        """(K, W) transit depth per sub-sample of exposure i."""
        t = self.exp_start_days[i] + self.sample_mid_points / 86400e3
        return transit_shape(t)[:, None] * self.depth0[None, :]

    # orbit of the synthetic planet for the device light curves (HD 209458 b-like: examples/...parameters.yml)
    ORBIT = dict(period=3.524746, sma_over_rs=8.81, eccentricity=0.0, inclination=86.71, periastron=0.0)
    LD = (0.800627, -0.757066, 0.897268, -0.384804)     # examples/...parameters.yml:26

    def device_depths(self, i):
        """The light-curve inputs of exposure i for the device (K orbit phases, the planet's spectrum,
        four limb-darkening coefficients) instead of the K x W matrix of planet_signal(i): a physical
        transit of the same depth spectrum (observation.py:293-357)."""
        from . import lightcurve
        t = self.exp_start_days[i] + self.sample_mid_points / 86400e3
        z_tr, hidden = lightcurve.depth_inputs(mid_time=0.0, time_array=t, rp_white=np.sqrt(self.depth0.mean()),
                                               **self.ORBIT)
        return lightcurve.DeviceDepths(z_tr, hidden, self.depth0, self.LD)

    def scale_factor(self, i):
        return 1.0 - 0.002 * np.exp(-i / 6.0)     # a hook-like visit trend

    def frame_kwargs(self, i, **override):
        """Keyword arguments for ExposureGenerator.scanning_frame of exposure i."""
        ssv = None
        if self.ssv:
            from .trend_generators.scan_speed_varations import SSVSine
            ssv = SSVSine(*self.ssv)
        # (the K x W depth matrix is only built when the caller does not bring its own planet_signal)
        signal = override["planet_signal"] if "planet_signal" in override else self.planet_signal(i)
        kw = dict(x_ref=self.x_refs[i], y_ref=self.y_refs[i], x_jitter=self.x_jitter, y_jitter=self.y_jitter,
                  wl=self.wl, stellar_flux=self.stellar_flux, planet_signal=signal,
                  scan_speed=self.scan_speed, sample_rate=10.0, sample_mid_points=self.sample_mid_points,
                  sample_durations=self.sample_durations, read_index=self.read_index, ssv_generator=ssv,
                  noise_mean=False, noise_std=False, add_dark=True, add_flat=True, cosmic_rate=self.cosmic_rate,
                  sky_background=self.sky[i], scale_factor=self.scale_factor(i), add_gain_variations=True,
                  add_non_linear=True, clip_values_det_limits=True, add_read_noise=True, add_stellar_noise=True,
                  add_initial_bias=True)
        kw.update(override)
        return kw
