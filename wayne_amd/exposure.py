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

    def generate_science_header(self):
        """Primary-header keywords describing the simulation (subset of exposure.py:216-410:
        the instrument / mode / simulation-switch keywords; no target ephemerides)."""
        e = self.exp_info
        cards = [("TELESCOP", "HST", ""), ("INSTRUME", "WFC3", ""), ("DETECTOR", "IR", ""),
                 ("SIM", True, "simulated exposure"), ("SIMULATR", "wayne_amd", "MI355X exposure synthesis"),
                 ("FILENAME", str(e.get("filename", "")), ""), ("OBSTYPE", str(e.get("OBSTYPE", "SPECTROSCOPIC")), ""),
                 ("FILTER", getattr(self.filter, "name", ""), ""),
                 ("NSAMP", int(e.get("NSAMP", 0)), ""), ("SAMP_SEQ", str(e.get("SAMPSEQ", "")), ""),
                 ("SUBARRAY", bool(e.get("SUBARRAY", 1024) != 1024), ""), ("SUBTYPE", "SQ%sSUB" % e.get("SUBARRAY", ""), ""),
                 ("EXPSTART", float(e.get("EXPSTART", 0.0)), "JD"), ("EXPEND", float(e.get("EXPEND", 0.0)), "JD"),
                 ("EXPTIME", float(e.get("EXPTIME", 0.0)), "seconds"),
                 ("SCAN", bool(e.get("SCAN", False)), ""), ("STAR-X", float(e.get("x_ref", 0.0)), ""),
                 ("STAR-Y", float(e.get("y_ref", 0.0)), ""), ("SAMPRATE", float(e.get("samp_rate", 0.0)), "ms"),
                 ("SIM-TIME", float(e.get("sim_time", 0.0)), "seconds to generate"),
                 ("NSE-MEAN", float(e.get("noise_mean") or 0.0), ""), ("NSE-STD", float(e.get("noise_std") or 0.0), ""),
                 ("ADD-DRK", bool(e.get("add_dark", False)), ""), ("ADD-FLAT", bool(e.get("add_flat", False)), ""),
                 ("ADD-GAIN", bool(e.get("add_gain", False)), ""), ("ADD-NLIN", bool(e.get("add_non_linear", False)), ""),
                 ("STAR-NSE", bool(e.get("add_stellar_noise", False)), ""),
                 ("CSMCRATE", float(e.get("cosmic_rate") if e.get("cosmic_rate") is not None else -1.0), ""),
                 ("SKY-LVL", float(e.get("sky_background") or 0.0), "ct/s"),
                 ("VSTTREND", float(e.get("scale_factor") if e.get("scale_factor") is not None else 1.0), ""),
                 ("CLIPVALS", bool(e.get("clip_values_det_limits", False)), "")]
        return fitsio.Header(cards)

    def generate_fits(self, out_dir="", filename=None, ldcoeffs=None):
        """Write the HST-style file (exposure.py:133-214): primary header, then for each
        read in REVERSE time order a float64 SCI image with SAMPNUM / SAMPTIME / DELTATIM /
        CRPIX1 followed by four data-less ERR, DQ, SAMP, TIME extensions, so that read r
        sits at HDU 1 + 5 (NSAMP - 1 - r) as in a real _raw/_ima file."""
        if filename is None:
            filename = self.exp_info.get("filename", "exposure_raw.fits")
        path = os.path.join(out_dir, filename)
        hdus = [fitsio.HDU(self.generate_science_header(), None)]
        n = len(self.reads)
        for i, (data, hdr) in enumerate(reversed(self.reads)):
            samp = n - 1 - i
            cards = [("SAMPNUM", samp, ""), ("SAMPTIME", float(hdr.get("SAMPTIME", 0.0)), "s"),
                     ("DELTATIM", float(hdr.get("DELTATIM", 0.0)), "s"), ("CRPIX1", hdr.get("CRPIX1", 0), ""),
                     ("EXTVER", i + 1, ""), ("BUNIT", "COUNTS", "")]
            hdus.append(fitsio.HDU(fitsio.Header(cards), np.asarray(data, dtype=np.float64), name="SCI"))
            for ext in ("ERR", "DQ", "SAMP", "TIME"):
                hdus.append(fitsio.HDU(fitsio.Header([("EXTVER", i + 1, "")]), None, name=ext))
        if os.path.exists(path):
            os.remove(path)
        fitsio.write(path, hdus)
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
