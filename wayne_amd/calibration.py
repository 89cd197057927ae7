"""Calibration planes for the exposure path.

The reference opens these from a downloaded calibration directory
(params.py:20-56) every time they are needed -- the gain and sky files once
per READ (detector.py:202, grism.py:417), the super-dark once per read
(detector.py:187).  Here they are loaded once into a CalibrationSet, cropped
to the sub-array once per mode, and uploaded to HBM once per context.

  CalibrationSet.synthetic(seed)       seeded stand-ins (the real files are not
                                       redistributable / not in this container):
                                       the recipe of SURVEY.md section 8(d)
  CalibrationSet.from_directory(path)  the reference's file names, read with
                                       wayne_amd.fitsio
"""
import os

import numpy as np

from . import fitsio
from .tools import crop_central_box

FLAT_FILES = {"G141": "WFC3.IR.G141.flat.2.fits", "G102": "WFC3.IR.G102.flat.2.fits"}        # grism.py:66-67, 453-454
SKY_FILES = {"G141": "WFC3.IR.G141.sky.V1.0.fits", "G102": "WFC3.IR.G102.sky.V1.0.fits"}      # grism.py:79-80, 457-458
SENS_FILES = {"G141": "WFC3.IR.G141.1st.sens.2.fits", "G102": "WFC3.IR.G102.1st.sens.2.fits"}  # grism.py:97-98, 467-468
PFL_FILE = "u4m1335mi_pfl.fits"    # detector.py:31
LIN_FILE = "u1k1727mi_lin.fits"    # detector.py:56-57


class CalibrationSet(object):
    def __init__(self):
        self.flat = {}        # grism -> (4, 1014, 1014) float32 cube f0..f3
        self.flat_wl = {}     # grism -> (WMIN, WMAX) angstrom
        self.sky = {}         # grism -> (1014, 1014) float32 master sky
        self.sens = {}        # grism -> (wl_um float64[n], val float64[n])
        self.pfl = None       # (1014, 1014) float32: pixel flat with the border cut (detector.py:203)
        self.lin = None       # (4, 1024, 1024) float32: c1..c4 (detector.py:58-67)
        self.dark = {}        # (SUBARRAY, SAMPSEQ) -> HDU list of the mode's super-dark (super_dark_hdus)
        self.dark_loader = None
        self.bias_256 = None  # (266, 266) float64

    # -- construction ----------------------------------------------------------
    @classmethod
    def synthetic(cls, seed=0, grisms=("G141", "G102")):
        """Seeded synthetic planes: flat cube f0 = 1 + N(0, 0.01),
        f1..f3 = N(0, 0.005), WMIN/WMAX = 10600/17000 A; pixel flat 1 + N(0, 0.01);
        master sky = smooth gradient of mean 1; linearity c2 = 7e-7 (1 + N(0, 0.05)),
        others 0 (about +5 % at 70 000 DN); super-dark 0.05 DN/s with error 0.02."""
        self = cls()
        rng = np.random.RandomState(seed)
        for g in grisms:
            cube = np.empty((4, 1014, 1014), dtype=np.float32)
            cube[0] = 1 + rng.normal(0, 0.01, (1014, 1014))
            for i in (1, 2, 3):
                cube[i] = rng.normal(0, 0.005, (1014, 1014))
            self.flat[g] = cube
            self.flat_wl[g] = (10600.0, 17000.0) if g == "G141" else (7800.0, 11800.0)
            yy, xx = np.mgrid[0:1014, 0:1014] / 1013.0
            sky = 1 + 0.08 * (xx - 0.5) + 0.05 * (yy - 0.5) + 0.03 * np.sin(2 * np.pi * xx) * np.cos(np.pi * yy)
            self.sky[g] = (sky / sky.mean()).astype(np.float32)
            c, w = (1.40, 0.33) if g == "G141" else (0.98, 0.19)
            wl = np.linspace(c - 0.5, c + 0.5, 201)
            self.sens[g] = (wl, 1e16 * np.exp(-((wl - c) / w) ** 4))
        self.pfl = (1 + rng.normal(0, 0.01, (1014, 1014))).astype(np.float32)
        lin = np.zeros((4, 1024, 1024), dtype=np.float32)
        lin[1] = 7e-7 * (1 + rng.normal(0, 0.05, (1024, 1024)))
        self.lin = lin
        self._synthetic_dark = True
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "wfc3_ir_initial_bias_256.npy")
        self.bias_256 = np.load(path)
        return self

    @classmethod
    def from_directory(cls, path, detector=None):
        """Load the files the reference expects in params._calb_dir."""
        self = cls()
        for g in ("G141", "G102"):
            fp = os.path.join(path, FLAT_FILES[g])
            if os.path.exists(fp):
                h = fitsio.read(fp)
                self.flat[g] = np.stack([np.asarray(h[i].data, dtype=np.float32) for i in range(4)])
                self.flat_wl[g] = (float(h[0].header["WMIN"]), float(h[0].header["WMAX"]))   # grism.py:71-72
            fp = os.path.join(path, SKY_FILES[g])
            if os.path.exists(fp):
                self.sky[g] = np.asarray(fitsio.read(fp)[0].data, dtype=np.float32)
            fp = os.path.join(path, SENS_FILES[g])
            if os.path.exists(fp):
                tbl = fitsio.read(fp)[1].data
                self.sens[g] = (np.asarray(tbl["WAVELENGTH"], dtype=np.float64) * 1e-4,       # A -> micron
                                np.asarray(tbl["SENSITIVITY"], dtype=np.float64))
        fp = os.path.join(path, PFL_FILE)
        if os.path.exists(fp):
            self.pfl = np.asarray(fitsio.read(fp)[1].data, dtype=np.float32)[5:-5, 5:-5]       # detector.py:203
        fp = os.path.join(path, LIN_FILE)
        if os.path.exists(fp):
            h = fitsio.read(fp)
            self.lin = np.stack([np.asarray(h[i].data, dtype=np.float32) for i in (1, 2, 3, 4)])
        self._synthetic_dark = False
        self._dir = path
        self._detector = detector
        bias = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "wfc3_ir_initial_bias_256.npy")
        self.bias_256 = np.load(bias)
        return self

    # -- access ------------------------------------------------------------------
    def sensitivity(self, grism):
        return self.sens[grism]

    def super_dark_hdus(self, subarray, sampseq, detector=None):
        """The mode's super-dark as the HDU list of the reference's file: [primary] + 5 extensions (SCI,
        ERR, DQ, SAMP, TIME) per read, last read first, so that read NSAMP-index n has its SCI frame at
        index -5 n and its error frame at -5 n + 1 (detector.py:183-190).  Entries are arrays or None.
        This is DATA ACCESS only -- which frame belongs to which read is decided by the caller.

        Synthetic: 0.05 DN/s with a +-20 % spatial pattern, error 0.02 with a +-50 % pattern and a
        sprinkling of zero / negative error pixels (the reference replaces those by 1e-5)."""
        from .detector import WFC3_IR
        det = detector or getattr(self, "_detector", None) or WFC3_IR()
        key = (subarray, sampseq)
        if key in self.dark:
            return self.dark[key]
        if self._synthetic_dark:
            times = det.modes_exp_table.get(subarray, {}).get(sampseq)
            if not times:
                det.dark_file(subarray, sampseq)                 # raises WFC3SimNoDarkFileError
                raise ValueError("no exposure-time table for the mode")
            S = min(subarray + 10, 1024)
            yy, xx = np.mgrid[0:S, 0:S]
            pat = (1.0 + 0.2 * np.sin(0.013 * yy + 0.5) * np.cos(0.017 * xx)).astype(np.float32)
            err = (0.02 * (1.0 + 0.5 * np.cos(0.011 * (xx + 2 * yy)))).astype(np.float32)
            err[(xx * 7 + yy * 13) % 997 == 0] = 0.0
            err[(xx * 5 + yy * 11) % 1999 == 0] = -0.01
            hdus = [None]
            for t in [0.0] + list(times[:15]):                   # zero read + 15 reads, stored last read first
                hdus[1:1] = [np.float32(0.05 * t) * pat, err, None, None, None]
        else:
            name = det.dark_file(subarray, sampseq)              # raises WFC3SimNoDarkFileError
            # 1 + 5 x 16 HDUs (SCI, ERR, DQ, SAMP, TIME per read, last read first): only SCI and ERR are ever used
            # (detector.py:185-190), so the other three of every read are dropped as they come in
            hdus = [h.data if (i == 0 or (i - 1) % 5 < 2) else None
                    for i, h in enumerate(fitsio.read(os.path.join(self._dir, name)))]
        self.dark[key] = hdus
        return hdus

    def dark_frames(self, subarray, sampseq, read_times, detector=None):
        """(sci, err), each (R, S, S) float32, for the R non-zero reads: read i (1-based NSAMP index
        i + 1) is HDU -5 (i + 1) of the mode's super-dark, its error the next HDU (detector.py:185-190)."""
        R = len(read_times)
        h = self.super_dark_hdus(subarray, sampseq, detector)
        sci = np.stack([np.asarray(h[-5 * (i + 1)], dtype=np.float32) for i in range(1, R + 1)])
        err = np.stack([np.asarray(h[-5 * (i + 1) + 1], dtype=np.float32) for i in range(1, R + 1)])
        return sci, err

    def for_mode(self, grism, subarray, sampseq, read_times, add_initial_bias=True, detector=None,
                 with_dark=True, flat_grism=None, flat_shift=0):
        """Planes centre-cropped to the sub-array, as Context.set_calibration takes them.
        `flat_grism`: take the flat cube of another grism (the reference flat-fields G102 exposures with
        the G141 cube, grism.py:428,453-454 -- `reference_quirks`).  `flat_shift`: roll the flat planes by
        that many pixels down / right, so that frame pixel (y, x) finds the flat of (y - shift, x - shift)
        with numpy's wrap-around: the reference's index offset (1014 - 1024) / 2 = -5 at the full array
        (grism.py:362-363), kept only with `reference_quirks`."""
        N = 1014 if subarray == 1024 else subarray
        S = N + 10
        out = {"subarray": subarray, "n_reads": len(read_times)}
        fg = flat_grism or grism
        if fg in self.flat:
            out["flat"] = [np.ascontiguousarray(crop_central_box(p, N)) for p in self.flat[fg]]      # grism.py:406-407
            if flat_shift:
                out["flat"] = [np.ascontiguousarray(np.roll(p, (flat_shift, flat_shift), axis=(0, 1))) for p in out["flat"]]
        if self.pfl is not None:
            out["pfl"] = np.ascontiguousarray(crop_central_box(self.pfl, N))                          # detector.py:206-207
        if grism in self.sky:
            out["sky"] = np.ascontiguousarray(crop_central_box(self.sky[grism], N))                   # grism.py:420-421
        if self.lin is not None:
            out["lin"] = [np.ascontiguousarray(crop_central_box(p, S)) for p in self.lin]             # detector.py:328-333
        if with_dark:
            from .detector import WFC3SimNoDarkFileError
            try:
                out["dark_sci"], out["dark_err"] = self.dark_frames(subarray, sampseq, read_times, detector)
            except WFC3SimNoDarkFileError:
                pass   # the caller switches the dark off with a warning (exposure_generator.py:417-423)
        if subarray == 256 and add_initial_bias and self.bias_256 is not None:
            out["zero_read"] = np.ascontiguousarray(self.bias_256, dtype=np.float64)                  # exposure_generator.py:456-458
        return out
