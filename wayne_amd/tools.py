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
