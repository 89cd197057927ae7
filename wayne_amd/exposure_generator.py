"""ExposureGenerator: builds one up-the-ramp WFC3-IR exposure.

Same class name, constructor and `scanning_frame` / `staring_frame` argument
lists as the reference (wayne/exposure_generator.py:16-58, 146-192), with
plain floats in fixed units instead of astropy Quantities:

    wl              micron            scan_speed      pixel / second
    sample_rate     millisecond       sample_mid_points / sample_durations   millisecond
    sky_background  counts / second   read times      second

What differs is where the work runs.  The reference loops over sub-samples on
the host (exposure_generator.py:336-394), calling the C thrower and several
full-frame numpy passes per sub-sample; here the host only prepares the small
K-vectors (sample timing, scan positions, jitter, SSV) and one descriptor, and
the whole exposure -- trace, counts, electron thrower, flat, sky, cosmic rays,
gain, dark, non-linearity, clipping, zero read, read noise -- is synthesised
by five HIP kernels through wayne_exposure_synthesize (include/wayne_hip.h).

Random numbers: the reference consumes one global MT19937 stream in exposure
order (run_visit.py:68-77); here every draw is a Philox counter keyed by
(seed, stage, exposure index, sub-sample | read, element), so exposures can be
generated in any order on any GPU with identical results.
"""
import time
import warnings

import numpy as np

from . import _lib, engine as _engine, exposure, lightcurve, tools
from .trend_generators import scan_speed_varations

MS_PER_YEAR = 365.25 * 86400. * 1000.
_CROP_CACHE = {}      # spectrum grid -> crop indices (build_descriptor)
_TIMES_CACHE = {}     # (read times, sample rate) -> sample starts / mid points / durations / read index


class WFC3SimNoDarkFileWarning(Warning):
    pass


class ExposureGenerator(object):
    def __init__(self, detector, grism, NSAMP, SAMPSEQ, SUBARRAY, planet=None,
                 filename="0001_raw.fits", start_JD=0.0, calibration=None, device=0, seed=0,
                 exposure_index=0):
        """:param calibration: wayne_amd.calibration.CalibrationSet (defaults to grism.calibration)
        :param device: GPU ordinal; :param seed: visit seed; :param exposure_index: extends
        every RNG counter so that exposures of a visit draw independent numbers."""
        self.detector, self.grism, self.planet = detector, grism, planet
        self.NSAMP, self.SAMPSEQ, self.SUBARRAY = NSAMP, SAMPSEQ, SUBARRAY
        self.calibration = calibration if calibration is not None else grism.calibration
        self.device, self.seed, self.exposure_index = device, seed, exposure_index
        self._submit_slot, self._pending = None, None      # pipelined use: submit() / collect()
        self._prepare_only, self._prepared = False, None   # ... or prepare() on one thread, launch() / collect() on another

        self.exptime = self.detector.exptime(NSAMP, SUBARRAY, SAMPSEQ)             # s
        self.read_times = self.detector.get_read_times(NSAMP, SUBARRAY, SAMPSEQ)   # s

        self.exp_info = {
            "filename": filename, "EXPSTART": start_JD, "EXPEND": start_JD + self.exptime / 86400.,
            "EXPTIME": self.exptime, "SCAN": False, "SCAN_DIR": None, "OBSTYPE": "SPECTROSCOPIC",
            "NSAMP": NSAMP, "SAMPSEQ": SAMPSEQ, "SUBARRAY": SUBARRAY, "samp_rate": 0.0, "sim_time": 0.0,
            "scan_speed_var": False, "noise_mean": False, "noise_std": False, "add_dark": False,
            "add_stellar_noise": False, "seed": seed,
        }

    # -- sample timing (host, K-vectors) ----------------------------------------
    def _gen_scanning_sample_times(self, sample_rate):
        """Sub-sample start / mid / duration (ms) and the index of the last
        sub-sample of each read (exposure_generator.py:531-579): sample at
        `sample_rate` from the previous read up to each read; the last sample
        before a read is cut short so that it ends on the read."""
        key = (tuple(float(t) for t in self.read_times), float(sample_rate))
        hit = _TIMES_CACHE.get(key)
        if hit is None:
            read_times = self.read_times * 1000.
            starts, read_index, i, previous = [], [], -1, 0.
            for read_time in read_times:
                s = np.arange(previous, read_time, sample_rate)
                starts.append(s)
                i += len(s)
                read_index.append(i)
                previous = read_time
            sample_starts = np.concatenate(starts)
            ends = np.roll(sample_starts, -1)
            ends[-1] = read_times[-1]
            sample_durations = ends - sample_starts
            sample_mid_points = sample_starts + (sample_durations / 2)
            if len(_TIMES_CACHE) > 16:
                _TIMES_CACHE.clear()
            hit = _TIMES_CACHE[key] = (sample_starts, sample_mid_points, sample_durations, read_index)
        # (the same numbers for every exposure of a visit; handed out as copies: callers scale the durations)
        return hit[0].copy(), hit[1].copy(), hit[2].copy(), list(hit[3])

    def _gen_sample_yref(self, y_ref, mid_points, scan_speed):
        """y of the star at each sub-sample mid-point; scan_speed in px/ms (:517-529)."""
        return y_ref + (np.asarray(mid_points, dtype=float) * scan_speed)

    # -- exposures ---------------------------------------------------------------
    def staring_frame(self, x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal,
                      sample_mid_points, sample_durations, read_index, noise_mean, noise_std, add_dark,
                      add_flat, cosmic_rate, sky_background, scale_factor, add_gain_variations,
                      add_non_linear, clip_values_det_limits, add_read_noise, add_stellar_noise,
                      add_initial_bias, progress_bar=None, threads=2, **kw):
        """A staring exposure is a scan at speed 0 sampled once per read (:146-176)."""
        self.scanning_frame(
            x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal, 0.0, MS_PER_YEAR,
            sample_mid_points, sample_durations, read_index, None, noise_mean, noise_std, add_dark, add_flat,
            cosmic_rate, sky_background, scale_factor, add_gain_variations, add_non_linear,
            clip_values_det_limits, add_read_noise, add_stellar_noise, add_initial_bias, progress_bar, threads,
            **kw)
        return self.exposure

    def scanning_frame(self, x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal,
                       scan_speed, sample_rate, sample_mid_points=None, sample_durations=None,
                       read_index=None, ssv_generator=None, noise_mean=False, noise_std=False,
                       add_dark=True, add_flat=True, cosmic_rate=None, sky_background=1.0,
                       scale_factor=None, add_gain_variations=True, add_non_linear=True,
                       clip_values_det_limits=True, add_read_noise=True, add_stellar_noise=True,
                       add_initial_bias=True, progress_bar=None, threads=2,
                       rng_mode=_lib.RNG_SPLIT, out_dtype=np.float32, reference_quirks=False,
                       record=None, exact_samplers=False):
        """Generate a spatially scanned exposure (exposure_generator.py:178-405).

        Extra keywords (not in the reference): `rng_mode` -- RNG_SPLIT (default:
        Philox-keyed streams, wide PSF component thrown per electron, narrow
        component drawn as one multinomial per bin), RNG_PHILOX (every electron
        thrown), or RNG_REPLAY (the reference's rand_r streams in the thrower,
        with `threads` selecting its OpenMP partition: bit-exact scatter); `out_dtype` -- float32 reads (the
        default HERE and in Observation, VisitRunner, the CLI and bench.py's `value`: ONE arithmetic for every entry
        point -- exact integer electron sums, then a float32 per-read chain; half the bytes to write and to carry
        over PCIe, <= 3 ulp of a float32 from the float64 result, no measurable effect on a recovered transit depth:
        profiles/r06/visit_science.json) or float64 reads (out_dtype=np.float64, CLI --float64-reads: what the
        reference's Exposure.reads hold, exposure.py:47,106-120 -- a DEVIATION of the default from the reference,
        chosen so that the benchmarked kernel is the one an API user gets; FITS files carry float64 SCI images
        either way); `reference_quirks` keeps the reference's -5 px frame
        offset at SUBARRAY=1024 (exposure_generator.py:630) and flat-fields G102 exposures with the
        G141 cube as the reference does (grism.py:428,453-454); `record`, if a dict,
        receives the device's intermediate products (counts, x, y per bin and
        sub-sample; electrons accumulated per read interval) for parity tests;
        `exact_samplers` evaluates the per-pixel Poisson / normal draws with IEEE
        divide / sqrt and accurate log / exp / sin / cos instead of the hardware
        approximations (same algorithm and streams; for parity runs).
        """
        start_time = time.time()
        slot = self._submit_slot
        eng = _engine.get_engine(self.device, self.grism, self.detector, self.calibration, self.NSAMP,
                                 self.SAMPSEQ, self.SUBARRAY, add_initial_bias, g102_flat_quirk=reference_quirks)
        desc = self.build_descriptor(
            eng, x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal, scan_speed, sample_rate,
            sample_mid_points, sample_durations, read_index, ssv_generator, noise_mean, noise_std, add_dark,
            add_flat, cosmic_rate, sky_background, scale_factor, add_gain_variations, add_non_linear,
            clip_values_det_limits, add_read_noise, add_stellar_noise, add_initial_bias, progress_bar, threads,
            rng_mode, out_dtype, reference_quirks, exact_samplers)
        if self._prepare_only:
            self._prepared = (eng, desc, start_time)        # host half done; launch(slot) does the rest
            return None
        if slot is not None:
            # pipelined use (submit / collect): enqueue everything and return; the reads are picked up later
            eng.ctx.upload(slot, desc)
            eng.ctx.run(slot)
            eng.ctx.fetch_async(slot)
            self._pending = (eng, slot, start_time)
            return None
        if record is None:
            reads = eng.ctx.synthesize(desc)
        else:
            eng.ctx.upload(0, desc)
            eng.ctx.run_front(0)
            record["counts"], record["x"], record["y"], record["acc"] = eng.ctx.debug_fetch(0, acc=True)
            record.update(self._host_vectors)
            eng.ctx.run_back(0)
            reads = eng.ctx.download(0)
        return self._fill_exposure(reads, start_time)

    def _fill_exposure(self, reads, start_time):
        # read 0 is the zero read (:301-303); reads 1..R carry their timing (:371-382)
        R = len(self.read_times)
        read_dt = self._read_dt
        self.exposure.add_read(reads[0], {"cumulative_exp_time": 0.0, "read_exp_time": 0.0, "CRPIX1": 0})
        for r in range(R):
            self.exposure.add_read(reads[r + 1], {"cumulative_exp_time": float(self.read_times[r]),
                                                  "read_exp_time": float(read_dt[r]), "CRPIX1": 0})
        assert len(self.exposure.reads) == self.NSAMP                                  # (:397)
        self.exp_info["sim_time"] = time.time() - start_time
        return self.exposure

    # -- pipelined generation: the host prepares exposure n+1 while the GPU works on n ---------------
    def submit(self, slot, *args, staring=False, **kw):
        """Enqueue a scanning (or staring) frame on context slot `slot` -- same arguments as
        scanning_frame / staring_frame -- and return at once; collect() returns the Exposure."""
        self._submit_slot = int(slot)
        try:
            (self.staring_frame if staring else self.scanning_frame)(*args, **kw)
        finally:
            self._submit_slot = None
        return self

    def prepare(self, *args, staring=False, **kw):
        """The HOST half of a scanning (or staring) frame -- same arguments -- and nothing else: sample timing, scan
        positions, jitter draws, the descriptor.  No GPU call, so it may run on another thread than the one that
        owns the context; launch(slot) then uploads and enqueues it, collect() returns the Exposure."""
        self._prepare_only = True
        try:
            (self.staring_frame if staring else self.scanning_frame)(*args, **kw)
        finally:
            self._prepare_only = False
        return self

    def launch(self, slot):
        """Upload a prepared frame into context slot `slot` and enqueue its kernels and the copy of its reads."""
        eng, desc, start_time = self._prepared
        self._prepared = None
        eng.ctx.upload(int(slot), desc)
        eng.ctx.run(int(slot))
        eng.ctx.fetch_async(int(slot))
        self._pending = (eng, int(slot), start_time)
        return self

    def collect(self):
        """Wait for a submitted frame -> Exposure (the reads are copied out of the slot's pinned buffer)."""
        eng, slot, start_time = self._pending
        self._pending = None
        reads = np.array(eng.ctx.wait(slot))
        return self._fill_exposure(reads, start_time)

    def build_descriptor(self, eng, x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal,
                         scan_speed, sample_rate, sample_mid_points=None, sample_durations=None,
                         read_index=None, ssv_generator=None, noise_mean=False, noise_std=False,
                         add_dark=True, add_flat=True, cosmic_rate=None, sky_background=1.0,
                         scale_factor=None, add_gain_variations=True, add_non_linear=True,
                         clip_values_det_limits=True, add_read_noise=True, add_stellar_noise=True,
                         add_initial_bias=True, progress_bar=None, threads=2,
                         rng_mode=_lib.RNG_SPLIT, out_dtype=np.float32, reference_quirks=False,
                         exact_samplers=False):
        """The host half of scanning_frame: sample timing, scan positions, SSV,
        jitter / seed draws, spectrum crop (exposure_generator.py:247-334) ->
        one wayne_exposure_desc for the device.  Pure host code (`eng` may be
        None when no GPU is involved, e.g. when sharding a visit on the CPU)."""
        wl = np.asarray(wl, dtype=float)
        stellar_flux = np.asarray(stellar_flux, dtype=float)
        scan_speed_ms = scan_speed / 1000.          # px/s -> px/ms (:247)

        if sample_mid_points is None and sample_durations is None and read_index is None:
            _, sample_mid_points, sample_durations, read_index = self._gen_scanning_sample_times(sample_rate)
        sample_mid_points = np.asarray(sample_mid_points, dtype=float)
        sample_durations = np.asarray(sample_durations, dtype=float)

        s_y_refs = self._gen_sample_yref(y_ref, sample_mid_points, scan_speed_ms)     # (:258)
        if ssv_generator is not None:
            if isinstance(ssv_generator, scan_speed_varations.SSVModulatedSine):          # (:263-267)
                # its scalar draws are keyed by (visit seed, exposure): order-independent
                ssv_generator.rng_seed = (self.seed * 1000003 + self.exposure_index * 7919 + 12345) & 0x7FFFFFFF
                sample_durations, read_index = ssv_generator.get_subsample_exposure_times(
                    s_y_refs, sample_durations, self.read_times, sample_rate)
                sample_durations = np.asarray(sample_durations, dtype=float)
                # it yields len(tt) durations; samples without one expose for 0 ms (:337-342)
                read_index = [min(int(b), len(sample_mid_points) - 1) for b in read_index]
                read_index[-1] = len(sample_mid_points) - 1
            else:
                sample_durations = np.asarray(ssv_generator.get_subsample_exposure_times(
                    s_y_refs, sample_durations, self.read_times, sample_rate), dtype=float)   # (:272-273)

        self.exp_info.update({
            "SCAN": True, "SCAN_DIR": 1, "samp_rate": sample_rate, "x_ref": x_ref, "y_ref": y_ref,
            "noise_mean": noise_mean, "noise_std": noise_std, "add_dark": add_dark, "add_flat": add_flat,
            "add_gain": add_gain_variations, "add_non_linear": add_non_linear,
            "add_stellar_noise": add_stellar_noise, "cosmic_rate": cosmic_rate,
            "sky_background": sky_background, "scale_factor": scale_factor,
            "clip_values_det_limits": clip_values_det_limits,
        })
        self.exposure = exposure.Exposure(self.detector, self.grism, self.planet, self.exp_info)

        if add_dark and eng is not None and not eng.has_dark:
            # the reference switches the dark off with a warning when the mode has no super-dark (:414-423)
            warnings.warn("No Dark file found for SAMPSEQ = {}, SUBARRAY={} - Switching Dark Off".format(
                self.SAMPSEQ, self.SUBARRAY), WFC3SimNoDarkFileWarning)
            add_dark = False
            self.exposure.exp_info["add_dark"] = False

        K = len(sample_mid_points)
        R = len(self.read_times)
        # per-exposure draws (:327-329): replay seeds and per-sub-sample jitter
        z_x, z_y, s_rand_seeds = _lib.host_sample_draws(self.seed, self.exposure_index, K)
        s_x = x_ref + z_x * x_jitter
        s_y_refs = np.asarray(s_y_refs, dtype=float)
        s_y = s_y_refs[np.minimum(np.arange(K), len(s_y_refs) - 1)] + z_y * y_jitter
        # a sub-sample without a duration (bad SSV) exposes for 0 ms (:337-342)
        s_dur = np.zeros(K)
        n_dur = min(K, len(sample_durations))
        s_dur[:n_dur] = np.asarray(sample_durations, dtype=float)[:n_dur]

        # crop to the grism's limits (:332-334); every exposure of a visit brings the same grid: looked up once
        key = (wl.size, float(wl[0]), float(wl[-1]), float(wl[wl.size // 2]), self.grism.wl_limits[0],
               self.grism.wl_limits[-1])
        hit = _CROP_CACHE.get(key)
        if hit is None or not np.array_equal(hit[0], wl):
            if len(_CROP_CACHE) > 16:
                _CROP_CACHE.clear()
            hit = _CROP_CACHE[key] = (wl.copy(), tools.crop_spectrum_ind(self.grism.wl_limits[0],
                                                                         self.grism.wl_limits[-1], wl))
        i0, i1 = hit[1]
        s_wl = wl[i0:i1]
        flux = stellar_flux[i0:i1]
        depth = None
        lc = {}
        if isinstance(planet_signal, lightcurve.DeviceDepths):
            # the K x W matrix is computed on the device from K + W + 4 numbers
            lc = dict(lc_z=planet_signal.z_tr, lc_hidden=planet_signal.hidden,
                      lc_rp=np.sqrt(planet_signal.planet_spectrum[i0:i1]), lc_ld=planet_signal.ld)
        elif planet_signal is not None:
            depth = np.ascontiguousarray(np.asarray(planet_signal, dtype=float)[:, i0:i1])

        # the read that closes each sub-sample (`if i in read_index`, :361)
        read_index = list(read_index)
        if len(read_index) != R or read_index[-1] != K - 1:
            raise ValueError("read_index must name the last sub-sample of each of the %d reads" % R)
        sample_read = np.searchsorted(np.asarray(read_index), np.arange(K), side="left").astype(np.int32)
        read_dt = np.diff(np.concatenate([[0.0], self.read_times]))                   # (:362-365)

        flags = 0
        for on, bit in ((add_flat, _lib.F_ADD_FLAT), (add_gain_variations, _lib.F_ADD_GAIN_VARIATIONS),
                        (add_non_linear, _lib.F_ADD_NON_LINEAR), (clip_values_det_limits, _lib.F_CLIP_DET_LIMITS),
                        (add_read_noise, _lib.F_ADD_READ_NOISE), (add_stellar_noise, _lib.F_ADD_STELLAR_NOISE),
                        (add_dark, _lib.F_ADD_DARK), (add_initial_bias, _lib.F_ADD_INITIAL_BIAS),
                        (np.dtype(out_dtype) == np.float64, _lib.F_OUT_F64),
                        (exact_samplers, _lib.F_EXACT_SAMPLERS)):
            if on:
                flags |= bit
        # frame offset 507 - SUBARRAY/2 (:630).  At 1024 the reference gets -5,
        # which shifts the spectrum by +5 px on a 1014 frame: use 0.
        sub_scale = 507 - self.SUBARRAY // 2
        if self.SUBARRAY == 1024 and not reference_quirks:
            sub_scale = 0

        if eng is not None:
            eng.check_descriptor(sub_scale)
        self._read_dt = read_dt
        self._host_vectors = {"x_ref": s_x, "y_ref": s_y, "dur": s_dur, "seeds": s_rand_seeds, "read": sample_read}
        return _lib.make_desc(
            self.seed, self.exposure_index, flags, sub_scale, s_wl, flux, depth, s_x, s_y, s_dur,
            sample_read, read_dt, replay_seed=s_rand_seeds, rng_mode=rng_mode, threads_compat=threads,
            sky_ct_s=float(sky_background) if sky_background else 0.0,
            cosmic_rate=-1.0 if cosmic_rate is None else float(cosmic_rate),
            scale_factor=1.0 if scale_factor is None else float(scale_factor),
            noise_mean=float(noise_mean) if noise_mean else 0.0,
            noise_std=float(noise_std) if noise_std else 0.0, **lc)

    def direct_image(self, x_ref, y_ref):
        """The unscaled 2-D gaussian direct image used to calibrate x_ref / y_ref
        (exposure_generator.py:83-144): zero read + one read, no detector effects."""
        self.exp_info.update({"OBSTYPE": "IMAGING", "x_ref": x_ref, "NSAMP": 2, "SAMP-SEQ": "RAPID",
                              "y_ref": y_ref, "add_flat": False, "add_gain": False, "add_non_linear": False,
                              "add_read_noise": False, "cosmic_rate": 0, "sky_background": 0.0,
                              "scale_factor": 1, "clip_values_det_limits": False})
        self.exposure = exposure.Exposure(self.detector, None, self.planet, self.exp_info)
        self.exposure.add_read(self.detector.gen_pixel_array(self.SUBARRAY, light_sensitive=False))
        n = self.SUBARRAY
        x, y = np.meshgrid(np.arange(n, dtype=float) + 0.5, np.arange(n, dtype=float) + 0.5)
        x0 = x_ref - (507.0 - n / 2.0)
        y0 = y_ref - (507.0 - n / 2.0)
        sigma = 2.0
        di = 10000.0 * np.exp(-((x0 - x) ** 2 + (y0 - y) ** 2) / (2.0 * sigma * sigma))
        self.exposure.add_read(di, {"read_exp_time": 0.0, "cumulative_exp_time": 0.0, "CRPIX1": -5})
        return self.exposure
