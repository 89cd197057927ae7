"""The Exposure container: the reads of one up-the-ramp exposure and their
HST-style FITS file.

Mirror of the reference's wayne/exposure.py.  In the reference this class also
DOES the post-ramp work read by read (apply_non_linear, add_dark_current,
scale_counts_between_limits, reset_reference_pixels, add_zero_read,
add_read_noise: exposure.py:49-131); here all of that has already happened in
the fused k_ramp kernel, and the class only holds the finished reads (read 0 =
zero read, exposure.py:47) and writes them out (exposure.py:133-214).
"""
import os
import sys
import queue
import threading

import numpy as np


from . import __version__ as _VERSION
from . import fitsio


class Exposure(object):
    def __init__(self, detector=None, filter=None, planet=None, exp_info=None):
        self.detector = detector
        self.filter = filter
        self.planet = planet
        self.exp_info = exp_info or {}
        self.SUBARRAY = self.exp_info.get("SUBARRAY")
        self.NSAMP = self.exp_info.get("NSAMP")
        self.SAMPSEQ = self.exp_info.get("SAMPSEQ")
        self.reads = []   # [(array (S, S), header dict)], read 0 first

    def add_read(self, data, read_info=None):
        self.reads.append((data, self.generate_read_header(read_info) if read_info is not None else {}))

    def generate_read_header(self, read_info):
        """SAMPTIME / DELTATIM / CRPIX1 of a read (exposure.py:412-429)."""
        return {"SAMPTIME": float(read_info.get("cumulative_exp_time", 0.0)),
                "DELTATIM": float(read_info.get("read_exp_time", 0.0)),
                "CRPIX1": read_info.get("CRPIX1", 0)}

    def generate_science_header(self, ldcoeffs=None):
        """The primary header of the reference's files, keyword for keyword (exposure.py:216-410): HST
        identification, target, exposure times as Modified Julian Dates, instrument configuration, the
        simulation switches, package versions and -- when a planet is attached -- its orbital elements
        and limb-darkening coefficients.  astropy / pandas versions are not written (neither is used)."""
        import datetime
        import platform
        e = self.exp_info
        planet = self.planet

        def blank(text=""):
            return ("", text, "")

        sub = e.get("SUBARRAY", 1024)
        cards = [
            ("DATE", datetime.datetime.now().strftime("%Y-%m-%d"), "date this file was written (yyyy-mm-dd)"),
            ("FILENAME", str(e.get("filename", "")), "name of file"),
            ("FILETYPE", "SCI", "type of data found in data file"),
            blank(),
            ("TELESCOP", getattr(self.detector, "telescope", "HST"), "telescope used to acquire data"),
            ("INSTRUME", getattr(self.detector, "instrument", "WFC3"), "identifier for instrument used to acquire data"),
            ("EQUINOX", 2000.0, "equinox of celestial coord. system"),
            blank(), blank("/ DATA DESCRIPTION KEYWORDS"), blank(),
            ("PRIMESI", getattr(self.detector, "instrument", "WFC3"), "instrument designated as prime"),
            blank(), blank("/ TARGET INFORMATION"), blank(),
            ("TARGNAME", str(getattr(planet, "name", "None")), "proposer's target name"),
            ("RA_TARG", float(getattr(planet, "ra_deg", 0.0) or 0.0), "right ascension of the target (deg) (J2000)"),
            ("DEC_TARG", float(getattr(planet, "dec_deg", 0.0) or 0.0), "declination of the target (deg) (J2000)"),
            blank(), blank("/ EXPOSURE INFORMATION"), blank(),
            ("DATE-OBS", False, "UT date of start of observation (yyyy-mm-dd)"),
            ("TIME-OBS", False, "UT time of start of observation (hh:mm:ss)"),
            ("EXPSTART", float(e.get("EXPSTART", 0.0)) - 2400000.5, "exposure start time (Modified Julian Date)"),
            ("EXPEND", float(e.get("EXPEND", 0.0)) - 2400000.5, "exposure end time (Modified Julian Date)"),
            ("EXPTIME", float(e.get("EXPTIME", 0.0)), "exposure duration (seconds)--calculated"),
            blank(), blank("/ TARGET OFFSETS (POSTARGS)"), blank(),
            ("POSTARG1", 0.0, "POSTARG in axis 1 direction"),
            ("POSTARG2", e.get("SCAN_DIR") if e.get("SCAN_DIR") is not None else 0, "POSTARG in axis 2 direction"),
            blank(), blank("/ INSTRUMENT CONFIGURATION INFORMATION"), blank(),
            ("OBSTYPE", str(e.get("OBSTYPE", "SPECTROSCOPIC")), "observation type - imaging or spectroscopic"),
            ("OBSMODE", "MULTIACCUM", "operating mode"),
            ("SCLAMP", "NONE", "lamp status, NONE or name of lamp which is on"),
            ("SUBARRAY", bool(sub != 1024), "data from a subarray (T) or full frame (F)"),
            ("SUBTYPE", "SQ%sSUB" % sub, ""),
            ("DETECTOR", getattr(self.detector, "detector_type", "IR"), "detector in use: UVIS or IR"),
            ("FILTER", getattr(self.filter, "name", ""), "element selected from filter wheel"),
            ("SAMP_SEQ", str(e.get("SAMPSEQ", "")), "MultiAccum exposure time sequence name"),
            ("NSAMP", int(e.get("NSAMP", 0)), "number of MULTIACCUM samples"),
            ("SAMPZERO", 0.0, "sample time of the zeroth read (sec)"),
            ("APERTURE", "GRISM%s" % sub, "aperture name"),
            ("PROPAPER", "", "proposed aperture name"),
            ("DIRIMAGE", "NONE", "direct image for grism or prism exposure"),
            blank(), blank("/ Wayne"), blank(),
            ("SIM", True, "Wayne Simulation (T/F)"),
            ("SIM-VER", "wayne_amd " + _VERSION, "WFC3Sim Version Used"),
            ("SIM-TIME", float(e.get("sim_time", 0.0)), "Wayne exposure generation time (s)"),
            blank(),
            ("X-REF", float(e.get("x_ref", 0.0)), "x position of star on frame (full frame))"),
            ("Y-REF", float(e.get("y_ref", 0.0)), "y position of star on frame (full frame))"),
            ("SAMPRATE", float(e.get("samp_rate", 0.0)) * 1e-3, "How often exposure is sampled (s)"),
            ("NSE-MEAN", e.get("noise_mean") if e.get("noise_mean") else False, "mean of normal noise (per s per pix)"),
            ("NSE-STD", e.get("noise_std") if e.get("noise_std") else False, "std of normal noise (per s per pix)"),
            ("ADD-DRK", bool(e.get("add_dark", False)), "dark current added (T/F)"),
            ("ADD-FLAT", bool(e.get("add_flat", False)), "flat field added (T/F)"),
            ("ADD-GAIN", bool(e.get("add_gain", False)), "gain variations added (T/F)"),
            ("ADD-NLIN", bool(e.get("add_non_linear", False)), "non-linearity effects added (T/F)"),
            ("STAR-NSE", bool(e.get("add_stellar_noise", False)), "Stellar Noise Added (T/F)"),
            ("CSMCRATE", float(e.get("cosmic_rate")) if e.get("cosmic_rate") is not None else False,
             "Rate of cosmic hits (per s)"),
            ("SKY-LVL", float(e.get("sky_background") or 0.0), "multiple of master sky per s"),
            ("VSTTREND", float(e.get("scale_factor")) if e.get("scale_factor") is not None else False,
             "visit trend scale factor"),
            ("CLIPVALS", bool(e.get("clip_values_det_limits", False)), "pixels clipped to detector range (T/F)"),
            ("RANDSEED", int(e.get("seed", 0)), "seed used for the visit"),
            ("SCAN", bool(e.get("SCAN", False)), "spatial scan (T/F); not a reference keyword"),
            blank(), blank("/ Wayne Package Versions Used"), blank(),
            ("V-PY", platform.python_version(), "Python version used"),
            ("V-NP", np.__version__, "NumPy version used"),
        ]
        try:
            import scipy
            cards.append(("V-SP", scipy.__version__, "SciPy version used"))
        except ImportError:
            pass
        if planet is not None and getattr(planet, "P", None) is not None:
            cards += [blank(), blank("/ Wayne Observation Parameters"), blank(),
                      ("MID-TRAN", float(planet.transittime) if planet.transittime is not None else False,
                       "Time of mid transit (JD)"),
                      ("PERIOD", float(planet.P), "Orbital Period (days)"),
                      ("SMA", float(planet.sma_over_rs) if planet.a and planet.Rs else False, "Semi-major axis (a/R_s)"),
                      ("INC", float(planet.i) if planet.i is not None else False, "Orbital Inclination (deg)"),
                      ("ECC", float(planet.e or 0.0), "Orbital Eccentricity"),
                      ("PERI", "%s" % planet.periastron, "Argument or periastron")]
            ld = ldcoeffs if ldcoeffs is not None else getattr(planet, "ldcoeffs", None)
            if ld is not None:
                for n_, v_ in enumerate(ld, 1):
                    cards.append(("LD%d" % n_, float(v_), "Non-linear limb darkening coeff %d" % n_))
        cards.append(("STARX", float(e.get("x_ref", 0.0)), "x position of star on frame (full frame))"))
        return fitsio.Header(cards)

    def generate_fits(self, out_dir="", filename=None, ldcoeffs=None):
        """Write the HST-style file (exposure.py:133-214): primary header, then for each
        read in REVERSE time order a float64 SCI image with SAMPNUM / SAMPTIME / DELTATIM /
        CRPIX1 followed by four data-less ERR, DQ, SAMP, TIME extensions, so that read r
        sits at HDU 1 + 5 (NSAMP - 1 - r) as in a real _raw/_ima file."""
        if filename is None:
            filename = self.exp_info.get("filename", "exposure_raw.fits")
        path = os.path.join(out_dir, filename)
        # Rendered piece by piece rather than HDU object by HDU object: the extension headers are the same bytes in every
        # file of a visit (cached blocks), the reads are byte-swapped into ONE big-endian cube and the file goes out in one
        # os.writev -- what is left under the interpreter lock per file is the primary header's ~100 (memoised) cards.
        # Byte for byte the file fitsio.write([HDU, ...]) produces (tests/test_visit_driver.py).
        primary = fitsio._image_hdu_parts(None, self.generate_science_header(ldcoeffs=ldcoeffs).cards, primary=True)[0]
        n = len(self.reads)
        arrs = [np.asarray(d) for d, _ in self.reads]
        same = n > 0 and all(a.shape == arrs[0].shape and a.ndim == 2 for a in arrs)
        pieces = [primary]
        if same:
            cube = np.empty((n,) + arrs[0].shape, dtype=">f8")
            for i, a in enumerate(arrs):
                cube[i] = a                                     # cast + byte swap, interpreter lock released
            pad = b"\x00" * ((-cube[0].nbytes) % fitsio.BLOCK)
        for i, (data, hdr) in enumerate(reversed(self.reads)):
            samp = n - 1 - i
            sampt, delt, crpix = float(hdr.get("SAMPTIME", 0.0)), float(hdr.get("DELTATIM", 0.0)), hdr.get("CRPIX1", 0)
            cards = [("SAMPNUM", samp, ""), ("SAMPTIME", sampt, "s"), ("DELTATIM", delt, "s"), ("CRPIX1", crpix, ""),
                     ("EXTVER", i + 1, ""), ("BUNIT", "COUNTS", "")]
            if same:
                shape = arrs[samp].shape
                pieces.append(fitsio.cached_header_block(("SCI", shape, samp, sampt, delt, type(crpix), crpix, i + 1),
                                                         cards, data_shape=shape, name="SCI"))
                pieces.append(memoryview(cube[samp].reshape(-1)).cast("B"))
                if pad:
                    pieces.append(pad)
            else:
                pieces += fitsio._image_hdu_parts(np.asarray(data, dtype=np.float64), cards, primary=False, name="SCI")
            for ext in ("ERR", "DQ", "SAMP", "TIME"):
                pieces.append(fitsio.cached_header_block((ext, i + 1), [("EXTVER", i + 1, "")], name=ext))
        fitsio.write_pieces(path, pieces)   # (replaces an existing file, as the reference's remove + writeto does: exposure.py:211-213)
        return path


class FitsWriterPool(object):
    """Write finished exposures to disk on background threads.

    Byte-swapping 16 float64 frames and writing the 134 MB file of a full-array exposure
    takes ~0.2 s on one core -- 200x longer than synthesising it on the GPU -- so a visit
    hands its exposures to this pool (numpy releases the GIL in astype / tobytes / write)
    and keeps generating.  `close()` waits for the queue to drain and re-raises a writer error."""

    def __init__(self, threads=None, max_pending=None):
        if threads is None:
            threads = int(os.environ.get("WAYNE_FITS_THREADS", "0")) or min(16, os.cpu_count() or 4)
        self._q = queue.Queue(maxsize=max_pending or 2 * threads)
        self._errors = []
        # the generating thread must get the interpreter back quickly after each (GIL-free) GPU call:
        # with the default 5 ms switch interval it waits behind the writers' header formatting
        self._switch_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(self._switch_interval, 2e-4))
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, threads))]
        for t in self._threads:
            t.start()

    def _work(self):
        while True:
            item = self._q.get()
            if item is None:
                self._q.task_done()
                return
            exposure, out_dir, filename = item
            try:
                exposure.generate_fits(out_dir, filename)
            except Exception as e:      # surfaced by close()
                self._errors.append(e)
            finally:
                self._q.task_done()

    def submit(self, exposure, out_dir, filename):
        """`exposure.reads` must own their data (not views of a buffer that will be reused)."""
        self._q.put((exposure, out_dir, filename))

    def close(self):
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        sys.setswitchinterval(self._switch_interval)
        if self._errors:
            raise self._errors[0]
