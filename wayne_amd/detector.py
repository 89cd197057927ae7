"""WFC3-IR detector description: mode timing tables and frame geometry.

Mirrors the part of the reference's wayne/detector.py that the exposure path
uses (detector.py:19-67 constants, :69-124 exptime / gen_pixel_array,
:211-267 read times and mode tables).  The per-pixel work that the reference
does in numpy here (gain, bias border, dark current, non-linearity, read
noise: detector.py:126-209, 318-350) runs inside the fused HIP ramp kernel
(wayne_amd/csrc/kernels.h, k_ramp); this class only carries the constants.
"""
import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class WFC3SimException(BaseException):
    pass


class WFC3SimSampleModeError(WFC3SimException):
    pass


class WFC3SimNoDarkFileError(WFC3SimException):
    pass


class WFC3_IR(object):
    """Constants and mode tables of the WFC3 IR channel."""

    def __init__(self):
        self.min_counts = -20          # DN  (detector.py:26)
        self.max_counts = 78000        # DN, 5 % non-linearity limit (detector.py:28)
        self.constant_gain = 2.35      # e-/DN (detector.py:30)
        self.read_noise = 14.1 / self.constant_gain  # DN (detector.py:33)
        self.telescope = "HST"
        self.instrument = "WFC3"
        self.detector_type = "IR"
        with open(os.path.join(_DATA, "wfc3_ir_modes.json")) as f:
            tables = json.load(f)
        # {SUBARRAY: {SAMPSEQ: [time of SAMPNUM 1, 2, ...]}}  (360 rows)
        self.modes_exp_table = {int(s): v for s, v in tables["exptime"].items()}
        # {SUBARRAY: {SAMPSEQ: super-dark file name}}          (19 rows)
        self.modes_calb_table = {int(s): v for s, v in tables["dark_file"].items()}
        self.initial_bias = os.path.join(_DATA, "wfc3_ir_initial_bias_256.npy")

    # -- mode tables ---------------------------------------------------------
    def _times(self, NSAMP, SUBARRAY, SAMPSEQ):
        try:
            times = self.modes_exp_table[SUBARRAY][SAMPSEQ]
        except (KeyError, TypeError):
            times = []
        try:
            sample_number = int(NSAMP) - 1   # the tables quote SAMPNUM = NSAMP - 1
        except (TypeError, ValueError):
            sample_number = 0
        if sample_number < 1 or sample_number > len(times):
            raise WFC3SimSampleModeError(
                "SAMPSEQ = {}, NSAMP={}, SUBARRAY={} is not a permitted combination".format(
                    SAMPSEQ, NSAMP, SUBARRAY))
        return times, sample_number

    def exptime(self, NSAMP, SUBARRAY, SAMPSEQ):
        """Total exposure time in seconds (detector.py:69-100)."""
        times, n = self._times(NSAMP, SUBARRAY, SAMPSEQ)
        return times[n - 1]

    def get_read_times(self, NSAMP, SUBARRAY, SAMPSEQ):
        """Time in seconds of each non-zero read (detector.py:211-246)."""
        if not 2 <= NSAMP <= 16:
            raise WFC3SimSampleModeError(
                "NSAMP must be an integer between 2 and 16, got {}".format(NSAMP))
        times, n = self._times(NSAMP, SUBARRAY, SAMPSEQ)
        return np.array(times[:n], dtype=float)

    def dark_file(self, SUBARRAY, SAMPSEQ):
        try:
            return self.modes_calb_table[SUBARRAY][SAMPSEQ]
        except KeyError:
            raise WFC3SimNoDarkFileError(
                "No Dark file found for SAMPSEQ = {}, SUBARRAY={}".format(SAMPSEQ, SUBARRAY))

    # -- geometry --------------------------------------------------------------
    @staticmethod
    def light_sensitive_size(subarray):
        return 1014 if subarray == 1024 else subarray      # detector.py:116-119

    @staticmethod
    def full_size(subarray):
        return min(subarray + 10, 1024)                    # detector.py:121-124

    def gen_pixel_array(self, subarray, light_sensitive=True):
        """Zero frame of the sub-array, with or without the 5-px border."""
        n = self.light_sensitive_size(subarray) if light_sensitive else self.full_size(subarray)
        return np.zeros((n, n))

    def num_exp_per_buffer(self, NSAMP, SUBARRAY):
        """Exposures before a buffer dump (detector.py:269-297)."""
        total_allowed_reads = min(2 * 16 * (1024 // SUBARRAY), 304)
        return int(np.floor(total_allowed_reads / (NSAMP + 1)))
