"""Small spectrum / array helpers used by the exposure path.

Same names and behaviour as the reference's wayne/tools.py for the four
helpers the hot path calls (tools.py:13-128, 317-324); everything
third-party-backed in that module (pysynphot rebinning, ephem HJD,
pylightcurve) is out of scope here.
"""
import numpy as np


def crop_spectrum_ind(min_wl, max_wl, wl):
    """Slice indices (imin, imax) of the samples with min_wl <= wl <= max_wl.

    `wl` must be increasing.  Reference: tools.py:46-77 (first sample at or
    above the lower limit, through the last sample at or below the upper one).
    """
    wl = np.asarray(wl, dtype=float)
    above = wl - min_wl
    above = np.where(above < 0, above.max(), above)
    imin = int(above.argmin())
    below = wl - max_wl
    below = np.where(below > 0, below.min(), below)
    imax = int(below.argmax()) + 1
    return imin, imax


def crop_spectrum(min_wl, max_wl, wl, flux):
    """(wl, flux) restricted to [min_wl, max_wl]; reference tools.py:13-44."""
    imin, imax = crop_spectrum_ind(min_wl, max_wl, wl)
    return wl[imin:imax], flux[imin:imax]


def _half_gaps(centers):
    centers = np.asarray(centers, dtype=float)
    gaps = np.empty_like(centers)
    gaps[1:] = (centers[1:] - centers[:-1]) / 2.
    gaps[0] = gaps[1]            # the first bin mirrors its only neighbour
    return centers, gaps


def bin_centers_to_edges(centers):
    """Bin edges half-way between neighbouring centres (tools.py:80-103)."""
    centers, gaps = _half_gaps(centers)
    edges = np.empty(len(centers) + 1)
    edges[:-1] = centers - gaps
    edges[-1] = centers[-1] + gaps[-1]
    return edges


def bin_centers_to_widths(centers):
    """Width of each bin: half-gap to the left plus half-gap to the right
    neighbour; end bins count their single gap twice (tools.py:106-128)."""
    centers, gaps = _half_gaps(centers)
    right = np.empty_like(gaps)
    right[:-1] = gaps[1:]
    right[-1] = gaps[-1]
    return gaps + right


def crop_central_box(array, size):
    """Central size x size box of a square array (tools.py:317-324).

    The reference slices [i:-i] with i = (len - size)/2 in Python-2 integer
    arithmetic, which returns an EMPTY array when len == size and a wrapped
    one when size > len (the SUBARRAY=1024 case).  Here
    equal sizes return the array itself and a larger size is an error.
    """
    n = len(array)
    if size == n:
        return array
    if size > n:
        raise ValueError("cannot crop a %d box out of a %d array" % (size, n))
    i = (n - size) // 2
    return array[i:n - i, i:n - i]


def sun_ra_dec(jd):
    """Apparent right ascension and declination of the Sun (radians, equinox of date) from the
    Astronomical Almanac's low-precision formulae (0.01 deg between 1950 and 2050) -- what the reference
    asks ephem for (tools.py:233-235)."""
    n = np.asarray(jd, dtype=float) - 2451545.0
    L = np.deg2rad((280.460 + 0.9856474 * n) % 360.0)           # mean longitude
    g = np.deg2rad((357.528 + 0.9856003 * n) % 360.0)           # mean anomaly
    lam = L + np.deg2rad(1.915) * np.sin(g) + np.deg2rad(0.020) * np.sin(2 * g)
    eps = np.deg2rad(23.439 - 0.0000004 * n)
    ra = np.arctan2(np.cos(eps) * np.sin(lam), np.cos(lam)) % (2 * np.pi)
    dec = np.arcsin(np.sin(eps) * np.sin(lam))
    return ra, dec


def jd_to_hjd(jd, ra_deg, dec_deg):
    """Julian date -> heliocentric Julian date for a target at (ra, dec) in degrees (tools.py:220-271:
    HJD = JD - (1 AU / c) [sin d sin d_sun + cos d cos d_sun cos(a - a_sun)] / 86400, with the Sun
    at exactly 1 AU as there).  Scalar or array."""
    ra, dec = np.deg2rad(ra_deg), np.deg2rad(dec_deg)
    ra_sun, dec_sun = sun_ra_dec(jd)
    a = 149597870700.0 / 299792458.0
    b = np.sin(dec) * np.sin(dec_sun)
    c = np.cos(dec) * np.cos(dec_sun) * np.cos(ra - ra_sun)
    return np.asarray(jd, dtype=float) - (a * (b + c)) / 86400.0


def detect_orbits(exp_start_times, separation=0.028):
    """Indices at which a new HST orbit starts: gaps of >= `separation` days
    (~40 min) between consecutive exposure starts (tools.py:274-300)."""
    t = np.asarray(exp_start_times, dtype=float)
    orbit_index = [0]
    for i in range(1, len(t)):
        if t[i] - t[i - 1] >= separation:
            orbit_index.append(i)
    return orbit_index


def wl_at_resolution(R, wl_min, wl_max):
    """Evenly spaced grid at resolution R about the mid wavelength (tools.py:303-314)."""
    mid_wl = (wl_max - wl_min) / 2 + wl_min
    delta_wl = mid_wl / R
    return np.arange(wl_min, wl_max + delta_wl, delta_wl)


def order_flux_grid(wavelength, spectrum):
    """Sort a spectrum by wavelength (tools.py:203-217)."""
    order = np.argsort(wavelength, kind="stable")
    return np.asarray(wavelength, dtype=float)[order], np.asarray(spectrum, dtype=float)[order]


def load_and_sort_spectrum(file_path):
    """Two-column text file: wavelength, flux or depth (tools.py:182-200)."""
    data = np.loadtxt(file_path)
    return order_flux_grid(data[:, 0], data[:, 1])


def load_pheonix_stellar_grid_fits(fits_file):
    """PHOENIX grid spectrum in a FITS binary table with columns Wavelength, Flux
    (tools.py:152-170): sorted, duplicate wavelengths removed."""
    from . import fitsio
    tab = fitsio.read(fits_file)[1].data
    names = {n.lower(): n for n in tab.dtype.names}
    wl, flux = order_flux_grid(tab[names["wavelength"]], tab[names["flux"]])
    keep = np.nonzero(np.diff(wl))
    return wl[keep], flux[keep]


def rebin_spec(wavelength, spectrum, new_wavelength):
    """Flux-conserving rebin onto the bins centred on `new_wavelength`.

    The reference delegates this to pysynphot (tools.py:131-149), which is not
    available: here the input spectrum, piecewise linear between its samples, is
    integrated exactly over each output bin (through its cumulative integral),
    the bin edges lying half-way between output centres.  Parity with pysynphot
    is unpinned; checked against oracle/wayne_oracle.py::rebin_spec."""
    wl = np.asarray(wavelength, dtype=float)
    sp = np.asarray(spectrum, dtype=float)
    new = np.asarray(new_wavelength, dtype=float)
    edges = bin_centers_to_edges(new)
    # integral of the piecewise-linear spectrum from wl[0] to each edge: the whole intervals below the edge plus
    # the trapezoid from the last sample to the edge (quadratic inside an interval); beyond the sampled range the
    # spectrum continues at its end values
    cum = np.concatenate([[0.0], np.cumsum(0.5 * (sp[1:] + sp[:-1]) * np.diff(wl))])
    j = np.clip(np.searchsorted(wl, edges, side="right") - 1, 0, wl.size - 1)
    f_edge = np.interp(edges, wl, sp)
    at_edges = cum[j] + 0.5 * (sp[j] + f_edge) * (edges - wl[j])
    below = edges < wl[0]
    at_edges[below] = sp[0] * (edges[below] - wl[0])
    return np.diff(at_edges) / np.diff(edges)


def blackbody_lambda(wl_um, T):
    """Planck B_lambda in erg / (s cm^2 A sr) at wavelengths in micron."""
    h, c, k = 6.62607015e-27, 2.99792458e10, 1.380649e-16      # cgs
    lam = np.asarray(wl_um, dtype=float) * 1e-4                  # cm
    b = 2 * h * c * c / lam ** 5 / np.expm1(h * c / (lam * k * T))   # per cm of wavelength
    return b * 1e-8                                              # per angstrom
