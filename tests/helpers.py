"""Shared builders for the exposure tests: the product objects and the oracle
objects over ONE calibration set and ONE synthetic visit."""
import numpy as np

from oracle import wayne_oracle as wo
from wayne_amd import calibration, detector, grism, synthetic
from wayne_amd.exposure_generator import ExposureGenerator

_cal = {}


def calibration_set(seed=11):
    if seed not in _cal:
        _cal[seed] = calibration.CalibrationSet.synthetic(seed)
    return _cal[seed]


def make_visit(name, n_exposures=1, seed=1963, **kw):
    cal = calibration_set()
    det = detector.WFC3_IR()
    gname = synthetic.CONFIGS[name]["grism"]
    gr = grism.G141(cal) if gname == "G141" else grism.G102(cal)
    return synthetic.Visit(name, det, gr, cal, n_exposures=n_exposures, seed=seed, **kw)


def product_generator(visit, i=0, device=0):
    return ExposureGenerator(visit.detector, visit.grism, visit.NSAMP, visit.SAMPSEQ, visit.SUBARRAY,
                             calibration=visit.calibration, device=device, seed=visit.seed, exposure_index=i)


def oracle_generator(visit):
    det, gr, eo = wo.from_calibration(visit.calibration, visit.grism.name, visit.NSAMP, visit.SAMPSEQ,
                                      visit.SUBARRAY)
    return eo


def oracle_kwargs(kw, seed=0, exposure=0):
    """scanning_frame keywords of the product -> the oracle's (same names).  `seed`, `exposure`: the visit seed and
    exposure index of the product generator -- the product keys the stream of a modulated-sine scan-speed generator
    by them (exposure_generator.py build_descriptor) where the reference draws from its global numpy stream
    (scan_speed_varations.py:100-167); the oracle's restatement gets a numpy legacy stream with that same seed."""
    from wayne_amd.trend_generators.scan_speed_varations import SSVModulatedSine
    kw = dict(kw)
    ssv = kw.get("ssv_generator")
    if isinstance(ssv, SSVModulatedSine):
        key = (int(seed) * 1000003 + int(exposure) * 7919 + 12345) & 0x7FFFFFFF
        kw["ssv_generator"] = wo.SSVModulatedSine(ssv.amplitude, ssv.period, ssv.blip_proba,
                                                  rs=np.random.RandomState(key))
    elif ssv is not None:
        kw["ssv_generator"] = wo.SSVSine(ssv.stddev, ssv.period, ssv.start_phase)
    return kw


def split_moved_bound(counts, total, exact=False):
    """Most electrons the device's split thrower may place in another pixel than oracle/split_oracle.c on the same
    counters (HISTORY.md section 6).  A binomial draw of a chain comes out differently when its uniform lands within
    the last bits of a step of the cdf / of a rejection test -- glibc against ocml with exact samplers, hardware
    rcp / exp / log in production math -- and the rest of THAT chain is then drawn afresh: a few sqrt(n) electrons
    of one bin or, where a group of 16 bins pools its rows, of one pooled column (n up to the group's electrons),
    whatever the total (scripts/diagnose_split_flip.py names such a draw: soak case 217, Binomial(8321, 0.1763),
    1 ulp of p).  So: a rate term for the many draws of a large input plus room for one flipped chain."""
    counts = np.asarray(counts, dtype=np.float64)
    n_chain = 16.0 * float(counts.max()) if counts.size else 0.0
    rate = 2e-6 if exact else 1e-4        # measured: 3.5e-7 / 5e-6 at 1e9 electrons, 2.5e-8 / 2.3e-5 at 1.2e8
    return 2 + rate * float(total) + 3.0 * np.sqrt(n_chain)


def reference_counts(v, kw, dur_ms):
    """The reference's counts chain, as written (exposure_generator.py:600-628, 678-684), per (sub-sample, cropped bin) BEFORE
    the Poisson draw / np.round:
        flux (1 - depth) x np.interp'ed sensitivity x tools.bin_centers_to_widths [um -> A: 1e4] x exptime [ms -> s: 1e-3] x scale
    evaluated in numpy from the visit's own arrays -- no oracle.  Returns (rates (K, W_cropped), cropped wavelengths)."""
    from wayne_amd import tools
    lo, hi = v.grism.wl_limits
    i0, i1 = tools.crop_spectrum_ind(lo, hi, v.wl.copy())
    wl, flux = v.wl[i0:i1], np.asarray(kw["stellar_flux"], dtype=float)[i0:i1]
    half = (wl - np.roll(wl, 1)) / 2.0                       # tools.py:106-128, spelled out
    half[0] = half[1]
    nxt = np.roll(half, -1)
    nxt[-1] = half[-1]
    dlam = half + nxt
    swl, sval = v.calibration.sensitivity(v.grism.name)
    sens = np.interp(wl, swl, sval)                           # grism.py:116-118
    depth = np.asarray(kw["planet_signal"])
    depth = depth[:, i0:i1] if depth.ndim == 2 else depth[i0:i1][None, :]
    dur = np.asarray(dur_ms, dtype=np.float64)[:, None]
    return flux[None, :] * (1.0 - depth) * sens[None, :] * dlam[None, :] * 1e4 * dur * 1e-3 * kw["scale_factor"], wl
