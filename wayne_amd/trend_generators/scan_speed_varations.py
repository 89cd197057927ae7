"""Scan-speed variations: per-sub-sample exposure-time scaling.

Mirror of wayne/trend_generators/scan_speed_varations.py:33-60 (SSVSine).
These are K-element host vectors that feed the device descriptor
(wayne_exposure_desc.dur_ms).  The reference's start_phase='rand' branch calls
a method that does not exist (`_flux_ssv_scaling`, :49) and cannot run; it is
rejected here.  SSVModulatedSine (:63-171) is not provided yet.
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
