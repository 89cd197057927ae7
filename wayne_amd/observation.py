"""Observation: a whole visit of exposures.

Same class and `setup_*` / `run_observation` methods as the reference's
wayne/observation.py:24-538, with plain floats (days, micron, seconds, px/s,
counts/s) instead of astropy / quantities objects and a small `Planet` record
instead of an exodata object.  What changed underneath:

  * light curves: the reference calls pylightcurve once per wavelength element
    per exposure (observation.py:349-355); here the K x W depth matrix of an
    exposure is computed on the GPU from the orbit (wayne_amd/lightcurve.py,
    k_lightcurve);
  * exposures are independent (counter-based RNG), so `run_observation` can
    take a (rank, world) pair and generate only its round-robin share.
"""
import collections
import os

import numpy as np

from . import lightcurve, tools
from .exposure_generator import ExposureGenerator
from .trend_generators import visit_trends
from .visit_planner import VisitPlanner

R_SUN_AU = 0.00465047      # solar radius in au


class Planet(object):
    """Orbital elements needed for the light curve (what the reference reads off an
    exodata Planet, observation.py:317-324)."""

    def __init__(self, name="planet", period=None, sma_au=None, stellar_radius_rsun=None, inclination=None,
                 eccentricity=0.0, periastron=0.0, transittime=None, rp_over_rs=None, star_temperature=None,
                 ra_deg=None, dec_deg=None):
        self.name = name
        self.ra_deg, self.dec_deg = ra_deg, dec_deg        # target coordinates (J2000, degrees) for JD -> HJD
        self.P, self.a, self.Rs = period, sma_au, stellar_radius_rsun
        self.i, self.e, self.periastron = inclination, eccentricity, periastron
        self.transittime = transittime
        self.rp_over_rs = rp_over_rs
        self.star_temperature = star_temperature

    @property
    def sma_over_rs(self):
        return self.a / (self.Rs * R_SUN_AU)


class Observation(object):
    def __init__(self, outdir="", calibration=None, device=0, seed=0):
        self.scanning = True
        self.outdir = outdir
        self.calibration = calibration
        self.device, self.seed = device, seed
        self._visit_trend = False
        self.ssv_gen = None
        self.noise_mean = self.noise_std = False
        # extra keywords for scanning_frame / staring_frame (rng_mode, out_dtype, exact_samplers, reference_quirks).
        # The visit driver asks for float32 reads: the FITS writer stores them as the reference's float64 SCI images
        # (BITPIX -64) either way, and a full-array exposure is 67 MB instead of 134 MB to bring over PCIe;
        # frame_options["out_dtype"] = np.float64 (CLI: --float64-reads) keeps the float64 arithmetic to the file.
        self.frame_options = {"out_dtype": np.float32}

    # -- setup_* (observation.py:46-291) ----------------------------------------
    def setup_observation(self, x_ref, y_ref, spatial_scan=False, scan_speed=False):
        self.x_ref, self.y_ref = x_ref, y_ref
        self.spatial_scan, self.scan_speed = spatial_scan, scan_speed

    def setup_simulator(self, sample_rate=False, clip_values_det_limits=True, threads=2):
        self.sample_rate = sample_rate
        self.clip_values_det_limits = clip_values_det_limits
        self.threads = threads

    def setup_target(self, planet, wavelengths, planet_spectrum, stellar_flux, transittime=None, ldcoeffs=None,
                     period=None, rp=None, sma=None, inclination=None, eccentricity=None, periastron=None,
                     stellar_radius=None):
        self.wl = np.asarray(wavelengths, dtype=float)
        self.stellar_flux = np.asarray(stellar_flux, dtype=float)
        self.planet_spectrum = None if planet_spectrum is None else np.asarray(planet_spectrum, dtype=float)
        assert len(self.wl) == len(self.stellar_flux)
        if not isinstance(planet, Planet):
            planet = Planet(name=str(planet))
        self.planet = planet
        if planet_spectrum is not None:
            assert len(self.wl) == len(self.planet_spectrum)
            self.transmission_spectroscopy = True
            for attr, val in (("P", period), ("a", sma), ("i", inclination), ("Rs", stellar_radius),
                              ("transittime", transittime)):
                if val:
                    setattr(planet, attr, val)
            if eccentricity or eccentricity == 0:
                planet.e = eccentricity
            if periastron or periastron == 0:
                planet.periastron = periastron
            if rp:
                planet.rp_over_rs = rp
            if not ldcoeffs:
                raise ValueError("ldcoeffs are required (the reference looks them up with pylightcurve.clablimb, "
                                 "tools.py:220-230, which is not available)")
            self.ldcoeffs = list(ldcoeffs)
            planet.ldcoeffs = self.ldcoeffs          # written into the FITS header (exposure.py:396-402)
        else:
            self.transmission_spectroscopy = False

    def setup_detector(self, detector, NSAMP, SAMPSEQ, SUBARRAY):
        self.detector, self.NSAMP, self.SAMPSEQ, self.SUBARRAY = detector, NSAMP, SAMPSEQ, SUBARRAY

    def setup_grism(self, grism):
        self.grism = grism
        if self.calibration is None:
            self.calibration = grism.calibration

    def setup_visit(self, start_JD, num_orbits, exp_start_times=False):
        self.start_JD, self.num_orbits = start_JD, num_orbits
        if exp_start_times is not False and exp_start_times is not None and len(np.atleast_1d(exp_start_times)):
            self.exp_start_times = np.asarray(exp_start_times, dtype=float)
            self.visit_plan = {"exp_start_times": self.exp_start_times,
                               "orbit_start_index": tools.detect_orbits(self.exp_start_times)}
        else:
            self.visit_plan = VisitPlanner(self.detector, self.NSAMP, self.SAMPSEQ, self.SUBARRAY, self.num_orbits,
                                           exp_overhead=3.0)            # observation.py:229-233
            self.exp_start_times = self.visit_plan["exp_times"] / (24. * 60.) + self.start_JD
            self.visit_plan["exp_start_times"] = self.exp_start_times

    def setup_reductions(self, add_dark=True, add_flat=True, add_gain_variations=True, add_non_linear=True,
                         add_initial_bias=True):
        self.add_dark, self.add_flat = add_dark, add_flat
        self.add_gain_variations, self.add_non_linear = add_gain_variations, add_non_linear
        self.add_initial_bias = add_initial_bias

    def setup_trends(self, ssv_gen, x_shifts=0, x_jitter=0.0000001, y_shifts=0, y_jitter=0.0000001):
        self.ssv_gen = ssv_gen
        self.x_shifts, self.x_jitter, self.y_shifts, self.y_jitter = x_shifts, x_jitter, y_shifts, y_jitter

    def setup_noise_sources(self, sky_background=1.0, cosmic_rate=11., add_read_noise=True, add_stellar_noise=True):
        self.sky_background, self.cosmic_rate = sky_background, cosmic_rate
        self.add_read_noise, self.add_stellar_noise = add_read_noise, add_stellar_noise

    def setup_gaussian_noise(self, noise_mean=False, noise_std=False):
        self.noise_mean, self.noise_std = noise_mean, noise_std

    def setup_visit_trend(self, visit_trend_coeffs):
        self._visit_trend = visit_trends.HookAndLongTermRamp(self.visit_plan, visit_trend_coeffs)

    # -- light curves ---------------------------------------------------------------
    def _orbit_args(self):
        p = self.planet
        W = p.periastron
        if W is None or (isinstance(W, float) and np.isnan(W)):
            W = 0.0
        return (float(p.P), float(p.sma_over_rs), float(p.e or 0.0), float(p.i), float(W), float(p.transittime))

    def generate_lightcurves(self, time_array, depth=False):
        """Normalised flux, shape (len(time_array), n_depths) (observation.py:293-357).
        The reference converts JD to HJD for the target's catalogue coordinates (observation.py:340,
        tools.py:220-271); here that happens when the planet carries ra_deg / dec_deg (there is no
        catalogue to look them up in), else times are used as given."""
        time_array = self._to_hjd(time_array)
        spectrum = np.array([depth]) if depth else self.planet_spectrum
        rp_white = self.planet.rp_over_rs or float(np.sqrt(np.mean(self.planet_spectrum)))
        z_tr, hidden = lightcurve.depth_inputs(*(self._orbit_args() + (time_array, rp_white)))
        return 1.0 - lightcurve.planet_depths(self.ldcoeffs, spectrum, z_tr, hidden)

    def device_depths(self, time_array):
        """The same per-sub-sample depths, as the recipe the GPU evaluates."""
        rp_white = self.planet.rp_over_rs or float(np.sqrt(np.mean(self.planet_spectrum)))
        z_tr, hidden = lightcurve.depth_inputs(*(self._orbit_args() + (self._to_hjd(time_array), rp_white)))
        return lightcurve.DeviceDepths(z_tr, hidden, self.planet_spectrum, self.ldcoeffs)

    def _to_hjd(self, time_array):
        p = self.planet
        if getattr(p, "ra_deg", None) is None or getattr(p, "dec_deg", None) is None:
            return time_array
        return tools.jd_to_hjd(time_array, p.ra_deg, p.dec_deg)

    def show_lightcurve(self):
        """White light curve of the planned visit -> (times, model); the reference also plots it."""
        t = self.exp_start_times
        if self.transmission_spectroscopy:
            depth = self.planet.rp_over_rs ** 2 if self.planet.rp_over_rs else float(np.mean(self.planet_spectrum))
            lc_model = self.generate_lightcurves(t, depth).T[0]
        else:
            lc_model = np.ones_like(t)
        if self._visit_trend:
            lc_model = np.asarray(self._visit_trend.scale_factors)[:len(lc_model)] * lc_model
        return t, lc_model

    # -- running ----------------------------------------------------------------------
    @staticmethod
    def _try_index(value, index):
        try:
            return value[index]
        except (TypeError, IndexError):
            return value

    def exposure_file_is_whole(self, number):
        """Is `NNNN_raw.fits` of exposure `number` (1-based) in the output directory, complete, and THIS visit's file?
        Files are written under a temporary name and renamed when finished (fitsio.write_pieces), so a file under its
        final name is whole unless something else truncated it: checked anyway -- the HDU structure is walked header by
        header (1 + 5 NSAMP HDUs ending exactly at the end of the file, SCI images of the mode's size) and the primary
        header must carry this exposure's start time and mode."""
        path = os.path.join(self.outdir, "{:04d}_raw.fits".format(number))
        if not os.path.isfile(path):
            return False
        from . import fitsio
        hdus = fitsio.scan(path)
        if hdus is None or len(hdus) != 1 + 5 * self.NSAMP:
            return False
        p0 = hdus[0][0]
        S = self.detector.frame_size(self.SUBARRAY) if hasattr(self.detector, "frame_size") else (
            1024 if self.SUBARRAY == 1024 else self.SUBARRAY + 10)
        try:
            same = (int(p0["NSAMP"]) == self.NSAMP and str(p0["SAMP_SEQ"]).strip() == self.SAMPSEQ and
                    abs(float(p0["EXPSTART"]) - (float(self.exp_start_times[number - 1]) - 2400000.5)) < 1e-7)
        except (KeyError, TypeError, ValueError):
            return False
        return bool(same) and all(size == S * S * 8 for (h, size) in hdus[1::5])

    def run_observation(self, rank=0, world=1, write_fits=True, resume=False):
        """Generate the direct image and every exposure (observation.py:388-413); with
        world > 1 only the exposures i = rank, rank + world, ... (round-robin sharding).
        `resume`: an exposure whose file is already in the output directory, whole and this visit's
        (exposure_file_is_whole), is not generated again -- what is left of a visit after a rank died is then only the
        files that are missing.  Every exposure's random streams are keyed by the visit seed and its own index, so the
        files of a resumed visit are those of an uninterrupted one (the reference, whose exposures share one global
        numpy stream, has no such restart: it deletes and rewrites, exposure.py:211-213)."""
        if write_fits and self.outdir and not os.path.exists(self.outdir):
            os.makedirs(self.outdir)
        frames = {}
        if write_fits and self.outdir:
            # a crash leaves at most half-written temporary files behind: this rank's are removed (never a final name)
            from . import fitsio
            for i in [-1] + list(range(rank, len(self.exp_start_times), world)):
                if i == -1 and rank != 0:
                    continue
                name = "0000_flt.fits" if i == -1 else "{:04d}_raw.fits".format(i + 1)
                part = os.path.join(self.outdir, name + fitsio.PART_SUFFIX)
                if os.path.exists(part):
                    os.remove(part)
        if rank == 0 and not (resume and write_fits and os.path.isfile(os.path.join(self.outdir, "0000_flt.fits"))
                              and self._fits_is_whole(os.path.join(self.outdir, "0000_flt.fits"))):
            frames[0] = self._generate_direct_image(write_fits)
        # files are written by background threads while the GPU works on the next exposures
        import sys
        from .exposure import FitsWriterPool
        old_interval = sys.getswitchinterval()
        pool = FitsWriterPool() if write_fits else None
        # ... and the host prepares exposure n+1.. while the GPU generates n: up to `depth` exposures
        # in flight on alternating context slots (even / odd slots run on different HIP streams)
        depth = 3
        in_flight = collections.deque()

        def finish_oldest():
            j, gen = in_flight.popleft()
            frame = gen.collect()
            if pool is not None:
                pool.submit(frame, self.outdir, "{:04d}_raw.fits".format(j + 1))
                frames[j + 1] = None          # on disk; do not keep 64 MB per exposure alive
            else:
                frames[j + 1] = frame

        # ... on two host threads: a producer runs the host half of every exposure (sample times, orbit phases, jitter
        # draws, the descriptor: ExposureGenerator.prepare -- no GPU call), this thread uploads, launches and collects.
        # Their C calls release the interpreter lock, so the example visit is paced by the device, not by Python.
        import queue
        import threading
        ahead = queue.Queue(maxsize=depth + 1)
        mine = list(range(rank, len(self.exp_start_times), world))
        self.skipped = []
        if resume and write_fits:
            self.skipped = [i for i in mine if self.exposure_file_is_whole(i + 1)]
            done = set(self.skipped)
            mine = [i for i in mine if i not in done]
        stop = threading.Event()            # set by this thread when it leaves the loop, for whatever reason
        # The context is created HERE, on the thread that will use it (upload / launch / collect): the producer's
        # prepare() then finds the engine in the cache instead of building the context, uploading grism and calibration,
        # on its own thread.
        from . import engine as _engine
        opts = dict(self.frame_options)
        _engine.get_engine(self.device, self.grism, self.detector, self.calibration, self.NSAMP, self.SAMPSEQ,
                           self.SUBARRAY, opts.get("add_initial_bias", self.add_initial_bias),
                           g102_flat_quirk=bool(opts.get("reference_quirks", False)))

        def put(item):
            """Queue.put that gives up when the consumer has gone (returns False)."""
            while not stop.is_set():
                try:
                    ahead.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for i in mine:
                    if stop.is_set():
                        return
                    if not put((i, self._generate_exposure(self.exp_start_times[i], i + 1, write_fits=False,
                                                           prepare_only=True))):
                        return
            except BaseException as e:          # surfaced in the consuming thread
                put(e)
                return
            put(None)

        producer = threading.Thread(target=produce, daemon=True)
        sys.setswitchinterval(min(old_interval, 2e-4))
        producer.start()
        try:
            n = 0
            while True:
                item = ahead.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                i, gen = item
                if len(in_flight) >= depth:
                    finish_oldest()
                in_flight.append((i, gen.launch(n % (depth + 1))))
                n += 1
            while in_flight:
                finish_oldest()
        finally:
            stop.set()                          # an error or Ctrl-C here: the producer stops after the exposure it is
            producer.join()                     # preparing, not after the rest of the visit's host work
            if pool is not None:
                pool.close()
            sys.setswitchinterval(old_interval)
        return frames

    @staticmethod
    def _fits_is_whole(path):
        from . import fitsio
        return fitsio.scan(path) is not None

    def _generate_exposure(self, expstart, number, write_fits=True, submit_slot=None, prepare_only=False):
        """observation.py:415-504.  With `submit_slot` the exposure is only enqueued on that context
        slot and the ExposureGenerator is returned: call its collect() for the Exposure.  With `prepare_only`
        only its host half runs (ExposureGenerator.prepare): launch(slot) and collect() follow on the context's thread."""
        index_number = number - 1
        filename = "{:04d}_raw.fits".format(number)
        exp_gen = ExposureGenerator(self.detector, self.grism, self.NSAMP, self.SAMPSEQ, self.SUBARRAY, self.planet,
                                    filename, expstart, calibration=self.calibration, device=self.device,
                                    seed=self.seed, exposure_index=index_number)
        sample_rate = self.sample_rate if self.spatial_scan else 365.25 * 86400. * 1000.
        _, sample_mid_points, sample_durations, read_index = exp_gen._gen_scanning_sample_times(sample_rate)
        time_array = expstart + sample_mid_points / (86400. * 1000.)
        planet_depths = self.device_depths(time_array) if self.transmission_spectroscopy else None
        x_ref = self._try_index(self.x_ref, index_number) + self.x_shifts * index_number
        y_ref = self._try_index(self.y_ref, index_number) + self.y_shifts * index_number
        sky_background = self._try_index(self.sky_background, index_number)
        scale_factor = self._visit_trend.get_scale_factor(index_number) if self._visit_trend else None
        common = dict(noise_mean=self.noise_mean, noise_std=self.noise_std, add_flat=self.add_flat,
                      add_dark=self.add_dark, scale_factor=scale_factor, sky_background=sky_background,
                      cosmic_rate=self.cosmic_rate, add_gain_variations=self.add_gain_variations,
                      add_non_linear=self.add_non_linear, clip_values_det_limits=self.clip_values_det_limits,
                      add_read_noise=self.add_read_noise, add_stellar_noise=self.add_stellar_noise,
                      add_initial_bias=self.add_initial_bias, threads=self.threads)
        common.update(self.frame_options)
        if self.spatial_scan:
            args = (x_ref, y_ref, self.x_jitter, self.y_jitter, self.wl, self.stellar_flux, planet_depths,
                    self.scan_speed, sample_rate, sample_mid_points, sample_durations, read_index)
            common["ssv_generator"] = self.ssv_gen
        else:
            args = (x_ref, y_ref, self.x_jitter, self.y_jitter, self.wl, self.stellar_flux, planet_depths,
                    sample_mid_points, sample_durations, read_index)
        if prepare_only:
            return exp_gen.prepare(*args, staring=not self.spatial_scan, **common)
        if submit_slot is not None:
            return exp_gen.submit(submit_slot, *args, staring=not self.spatial_scan, **common)
        exp_frame = exp_gen.scanning_frame(*args, **common) if self.spatial_scan else exp_gen.staring_frame(*args, **common)
        if write_fits:
            exp_frame.generate_fits(self.outdir, filename)
        return exp_frame

    def _generate_direct_image(self, write_fits=True):
        """observation.py:516-538."""
        di_start_JD = self.exp_start_times[0] - 1.0 / (24. * 60.)
        gen = ExposureGenerator(self.detector, self.grism, self.NSAMP, self.SAMPSEQ, self.SUBARRAY, self.planet,
                                "0000_flt.fits", di_start_JD, calibration=self.calibration, device=self.device)
        exp = gen.direct_image(self._try_index(self.x_ref, 0), self._try_index(self.y_ref, 0))
        if write_fits:
            exp.generate_fits(self.outdir, "0000_flt.fits")
        return exp
