"""One GPU's exposure engine: a wayne_ctx configured for a grism and a
(SUBARRAY, SAMPSEQ, NSAMP) mode, with calibration planes resident in HBM.

The reference re-derives this state for every exposure -- re-opening gain,
sky and dark FITS files per read (detector.py:187,202; grism.py:417).  Here it
is built once per (device, grism, mode) and reused for the whole visit.
"""
import numpy as np

from . import _lib

_engines = {}


class Engine(object):
    def __init__(self, device, grism, detector, calibration, NSAMP, SAMPSEQ, SUBARRAY,
                 add_initial_bias=True, g102_flat_quirk=False, flat_shift=0):
        self.device = device
        self.grism, self.detector, self.calibration = grism, detector, calibration
        self.NSAMP, self.SAMPSEQ, self.SUBARRAY = NSAMP, SAMPSEQ, SUBARRAY
        self.read_times = detector.get_read_times(NSAMP, SUBARRAY, SAMPSEQ)    # seconds
        self.flat_shift = int(flat_shift)      # +5 px roll of the flat planes that goes with sub_scale = -5 (reference_quirks)
        self.ctx = _lib.Context(device)
        sens_wl, sens_val = calibration.sensitivity(grism.name)
        # the reference's G102 inherits the G141 flat cube and its WMIN / WMAX (grism.py:428: G102.__init__
        # calls G141.__init__, which loads them; only the *path* is overridden afterwards, :453-454)
        flat_grism = "G141" if (g102_flat_quirk and grism.name == "G102") else grism.name
        wmin, wmax = calibration.flat_wl.get(flat_grism, (0.0, 1.0))
        self.ctx.set_grism(grism.trace_coeff, grism.wl_solution, grism.psf_ratio_poly.coeffs,
                           grism.psf_sigmal_poly.coeffs, grism.psf_sigmah_poly.coeffs,
                           sens_wl, sens_val, wmin, wmax)
        planes = calibration.for_mode(grism.name, SUBARRAY, SAMPSEQ, self.read_times,
                                      add_initial_bias=add_initial_bias, detector=detector, flat_grism=flat_grism,
                                      flat_shift=flat_shift)
        self.has_dark = "dark_sci" in planes
        self.ctx.set_calibration(planes["subarray"], planes["n_reads"], flat=planes.get("flat"),
                                 pfl=planes.get("pfl"), sky=planes.get("sky"), lin=planes.get("lin"),
                                 dark_sci=planes.get("dark_sci"), dark_err=planes.get("dark_err"),
                                 zero_read=planes.get("zero_read"))
        self.N, self.S, self.R = self.ctx.N, self.ctx.S, self.ctx.R

    def check_descriptor(self, sub_scale):
        """The descriptor's frame offset also indexes the flat (wayne_hip.hip: flat_off = sub_scale), so it must be
        the offset this engine's flat planes were laid out for: at SUBARRAY 1024 that is -5 with the planes rolled
        by +5 px (reference_quirks) or 0 with the planes as they are -- never a mix of the two."""
        if self.SUBARRAY == 1024 and int(sub_scale) != -self.flat_shift:
            raise ValueError("descriptor sub_scale %d does not go with this engine's flat planes (rolled by %d px): "
                             "build the engine and the descriptor with the same reference_quirks" % (
                                 sub_scale, self.flat_shift))

    def close(self):
        self.ctx.close()


def get_engine(device, grism, detector, calibration, NSAMP, SAMPSEQ, SUBARRAY, add_initial_bias=True,
               g102_flat_quirk=False):
    """Cached engine per (device, grism, mode, calibration object).  `g102_flat_quirk` (the caller's
    reference_quirks) selects the reference's flat handling: the G141 cube for G102 exposures, and at
    SUBARRAY = 1024 its -5 px flat index offset (grism.py:362-363)."""
    quirk = bool(g102_flat_quirk) and grism.name == "G102"
    shift = 5 if (bool(g102_flat_quirk) and SUBARRAY == 1024) else 0
    key = (device, grism.name, NSAMP, SAMPSEQ, SUBARRAY, id(calibration), bool(add_initial_bias), quirk, shift)
    eng = _engines.get(key)
    if eng is None:
        eng = Engine(device, grism, detector, calibration, NSAMP, SAMPSEQ, SUBARRAY, add_initial_bias, quirk, shift)
        _engines[key] = eng
    return eng


def close_all():
    for e in _engines.values():
        e.close()
    _engines.clear()
