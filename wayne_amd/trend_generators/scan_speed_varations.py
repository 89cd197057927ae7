"""Scan-speed variations: per-sub-sample exposure-time scaling.

Mirror of wayne/trend_generators/scan_speed_varations.py:33-60 (SSVSine).
These are K-element host vectors that feed the device descriptor
(wayne_exposure_desc.dur_ms).  The reference's start_phase='rand' branch calls
a method that does not exist (`_flux_ssv_scaling`, :49) and cannot run; it is
rejected here.
"""
import numpy as np


class SSVSine(object):
    def __init__(self, stddev=1.5, period=0.7, start_phase=0.0):
        if isinstance(start_phase, str):
            raise ValueError("start_phase='rand' is broken in the reference (scan_speed_varations.py:49); "
                             "pass a phase in radians")
        self.stddev = stddev
        self.period = period
        self.start_phase = start_phase

    def get_subsample_exposure_times(self, y_mid_points, sample_durations, subsample_exptime=None,
                                     total_exptime=None):
        """durations * (1 + stddev/100 * sin(period * (y - y_0) + phase))."""
        y = np.asarray(y_mid_points, dtype=float)
        zeroed_y_mid = y - y[0]
        ssv_scaling = (self.stddev / 100.) * np.sin((self.period * zeroed_y_mid) + self.start_phase) + 1.
        return np.asarray(sample_durations, dtype=float) * ssv_scaling


class SSVModulatedSine(object):
    """Sub-sample exposure times with a sine of slowly varying amplitude and
    period, an optional gaussian "blip", and the total exposure time and every
    read time preserved to the microsecond.

    Mirror of wayne/trend_generators/scan_speed_varations.py:63-171 (the
    reference notes it "isnt really implemented well": mid-points and scan
    positions are not updated, and it yields one sample fewer).  The reference
    draws ~10 + (a few hundred) scalars from the global numpy stream; here they
    come from a numpy legacy generator seeded per call by `rng_seed` (the
    exposure generator passes a value derived from (visit seed, exposure
    index)), so an exposure's times do not depend on which exposures ran before.
    Times in ms in, ms out; returns (durations_ms, read_indexes)."""

    def __init__(self, amplitude=10, period=1.1, blip_proba=1, rng_seed=0):
        self.amplitude = amplitude
        self.period = period
        self.blip_proba = blip_proba
        self.rng_seed = rng_seed

    def get_subsample_exposure_times(self, y_mid_points, sample_durations, read_times, sample_rate):
        """read_times in seconds, sample_rate in ms (as ExposureGenerator passes them)."""
        rs = np.random.RandomState(int(self.rng_seed) & 0x7FFFFFFF)
        read_times = np.asarray(read_times, dtype=float)
        rate = float(sample_rate) / 1000.0                      # seconds
        exptime = np.round(read_times[-1], 6)
        tt = np.arange(0, exptime, rate)
        n = len(tt)

        def slow_sine():
            return rs.normal(0.1, 0.05) * np.sin((2 * np.pi / rs.normal(2.0 * exptime, 0.5 * exptime)) * tt +
                                                 rs.random_sample() * 2 * np.pi)
        amp = 1.0 + slow_sine()
        if 100.0 * rs.random_sample() < self.blip_proba:
            amp = amp + rs.normal(1.0, 0.1) * np.exp(-(tt - rs.random_sample() * exptime) ** 2 /
                                                     (2 * (self.period / 2) ** 2))
        final_amp = rate * (self.amplitude / 100.0) * amp
        final_per = self.period * (1.0 + slow_sine())
        phase = rs.random_sample() * 2 * np.pi
        sub = np.round(rate + final_amp * np.sin((2 * np.pi / final_per) * tt + phase), 6)

        def spread(diff_us, lo, hi, sign):
            """add `sign` microseconds to |diff_us| random samples of [lo, hi)"""
            for _ in range(abs(diff_us)):
                sub[rs.randint(lo, hi)] += sign * 0.000001

        # total exposure time
        d = int(1e6 * np.round(exptime - np.sum(sub), 6))
        spread(d, 0, n, 1 if d > 0 else -1)
        breaks = [int(np.argmin(np.abs(np.cumsum(sub) - t))) for t in read_times]
        # first read: move microseconds between the first interval and the rest
        d = int(1e6 * np.round(read_times[0] - np.sum(sub[:breaks[0] + 1]), 6))
        sign = 1 if d > 0 else -1
        for i in np.int_(rs.power(3, abs(d)) * (breaks[0] + 1)):
            sub[breaks[0] - i] += sign * 0.000001
            if breaks[0] + 1 < n:
                sub[rs.randint(breaks[0] + 1, n)] -= sign * 0.000001
        # the other reads but the last
        for r in range(1, len(read_times) - 1):
            d = int(1e6 * np.round(read_times[r] - np.sum(sub[:breaks[r] + 1]), 6))
            sign = 1 if d > 0 else -1
            for _ in range(abs(d)):
                if breaks[r] > breaks[r - 1]:
                    sub[rs.randint(breaks[r - 1] + 1, breaks[r] + 1)] += sign * 0.000001
                if breaks[r] + 1 < n:
                    sub[rs.randint(breaks[r] + 1, n)] -= sign * 0.000001
        return sub * 1000.0, breaks
