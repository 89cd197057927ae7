"""oracle/wayne_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's exposure-synthesis path, rows A5-A16 of
SURVEY.md section 8(a).  Every function cites the reference lines it follows
(paths under the ucl-exoplanets/wayne tree).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.

Structure follows the reference on purpose -- one C thrower call plus several
full-frame numpy passes PER SUB-SAMPLE, per-read and post-ramp stages as
separate passes -- so that timing it is a fair "port" CPU baseline.

Pinning
  * trace / wavelength solution (A6, A7): tests/test_grism.py:16-51, 66-80
  * mode tables (A12 inputs):             tests/test_detector.py:16-32, 72-81
  * bin widths (A9):                      tests/test_tools.py:48-81
  * thrower (A1-A4):                      oracle/_ref (the reference's C compiled
                                          unmodified) and tests/golden/psf_*.npz
  * everything else (flat A11, counts chain A9, per-read A13, post-ramp A15):
    PARITY UNPINNED -- the reference has no test or fixture for these and its
    Python cannot be imported here (Python 2 only; astropy, pysynphot, ... absent;
    import-time download in params.py:41-56).  They are restated line against
    line, including numpy's float32 behaviour where the reference's arrays are
    float32, and reviewed against the citations.

Units are plain floats: wl micron, times ms unless a name says _s, scan speed
px/ms, sky counts/s.

Random numbers.  `LegacyDraws` draws from numpy's legacy MT19937 generator in
exactly the reference's call order (this is what the reference computes; the
stream depends on exposure order and cannot be sharded).  `PhiloxDraws`
mirrors the counter layout of the MI355X path (wayne_amd/csrc/philox.h) so
that device and oracle can be compared under a fixed seed.
"""
import json
import os

import numpy as np

from . import clib

HERE = os.path.dirname(os.path.abspath(__file__))
_DATA = os.path.join(os.path.dirname(HERE), "wayne_amd", "data")   # data tables only (json / npy)

# ---------------------------------------------------------------------------
# tools.py
# ---------------------------------------------------------------------------


def crop_spectrum_ind(min_wl, max_wl, wl):
    """tools.py:46-77."""
    wl_min_nearest = wl - min_wl
    wl_min_nearest[wl_min_nearest < 0] = wl_min_nearest.max()
    imin = wl_min_nearest.argmin()
    wl_max_nearest = wl - max_wl
    wl_max_nearest[wl_max_nearest > 0] = wl_max_nearest.min()
    imax = wl_max_nearest.argmax() + 1
    return int(imin), int(imax)


def crop_spectrum(min_wl, max_wl, wl, flux):
    """tools.py:13-44."""
    imin, imax = crop_spectrum_ind(min_wl, max_wl, wl)
    return wl[imin:imax], flux[imin:imax]


def bin_centers_to_edges(centers):
    """tools.py:80-103."""
    centers = np.asarray(centers, dtype=float)
    bin_range = (centers - np.roll(centers, 1)) / 2.
    bin_range[0] = bin_range[1]
    bin_edges = np.zeros(len(centers) + 1)
    bin_edges[:-1] = centers - bin_range
    bin_edges[-1] = centers[-1] + bin_range[-1]
    return bin_edges


def bin_centers_to_widths(centers):
    """tools.py:106-128."""
    centers = np.asarray(centers, dtype=float)
    bin_range = (centers - np.roll(centers, 1)) / 2.
    bin_range[0] = bin_range[1]
    bin_range_roll = np.roll(bin_range, -1)
    bin_range_roll[-1] = bin_range[-1]
    return bin_range + bin_range_roll


def crop_central_box(array, size):
    """tools.py:317-324, Python-2 integer division.  When len(array) == size
    the reference's array[0:-0] is EMPTY: identity here,
    the documented deviation for the 1014 frame."""
    if len(array) == size:
        return array
    index = (len(array) - size) // 2
    return array[index:-index, index:-index]


def rebin_spec(wavelength, spectrum, new_wavelength):
    """tools.py:131-149 hands this to pysynphot (an un-vendored dependency, version unpinned in setup.py:32-33, not
    installed here): `Observation(spec, flat filter, binset=new_wavelength, force='taper').binflux`.  pysynphot's
    published binning (observation.py, `initbinflux`): bin edges half-way between the binset wavelengths, the end
    bins as wide as their neighbours' half-spacing allows; the spectrum, piecewise linear between its samples, is
    integrated over each bin and divided by the bin's width.  Restated here bin by bin with an explicit merged
    grid and the trapezoid rule -- deliberately not the product's cumulative-integral interpolation.  PARITY
    UNPINNED against pysynphot itself."""
    wl = np.asarray(wavelength, dtype=float)
    sp = np.asarray(spectrum, dtype=float)
    new = np.asarray(new_wavelength, dtype=float)
    edges = bin_centers_to_edges(new)
    out = np.empty(new.size)
    for i in range(new.size):
        lo, hi = edges[i], edges[i + 1]
        inside = wl[(wl > lo) & (wl < hi)]
        grid = np.concatenate([[lo], inside, [hi]])
        # outside the sampled range the spectrum continues at its end values (np.interp's clamp)
        flux = np.interp(grid, wl, sp)
        area = 0.0
        for j in range(grid.size - 1):
            area += 0.5 * (flux[j] + flux[j + 1]) * (grid[j + 1] - grid[j])
        out[i] = area / (hi - lo)
    return out


# ---------------------------------------------------------------------------
# grism.py
# ---------------------------------------------------------------------------
G141_TRACE = (1.96882, 9.09159E-5, -1.93260E-3, 1.04275E-2, -7.96978E-6, -2.49607E-6, 1.45963E-9,
              1.39757E-8, 4.8494E-10)                                  # grism.py:756-757
G102_TRACE = (-3.55018E-1, 3.28722E-5, -1.44571E-3, 1.42852E-2, -7.20713E-6, -2.42542E-6, 1.18294E-9,
              1.19634E-8, 6.17274E-10)                                 # grism.py:762-764
G141_WLSOL = (8.95431E3, 9.35925E-2, 0, 4.51423E1, 3.17239E-4, 2.17055E-3, -7.42504E-7, 3.48639E-7,
              3.09213E-7)                                              # grism.py:768-769
G102_WLSOL = (6.38738E3, 4.55507E-2, 0, 2.35716E1, 3.60396E-4, 1.58739E-3, -4.25234E-7, -6.53726E-8,
              0.)                                                      # grism.py:773-774


def wavelength_calibration_coeffs(x_ref, y_ref, trace_coeff, wl_sol_coeff):
    """grism.py:779-803."""
    a, b = trace_coeff, wl_sol_coeff
    m_t = a[3] + a[4] * x_ref + a[5] * y_ref + a[6] * x_ref ** 2 + a[7] * x_ref * y_ref + a[8] * y_ref ** 2
    c_t = a[0] + a[1] * x_ref + a[2] * y_ref
    m_w = b[3] + b[4] * x_ref + b[5] * y_ref + b[6] * x_ref ** 2 + b[7] * x_ref * y_ref + b[8] * y_ref ** 2
    c_w = (b[0] + b[1] * x_ref) + b[2] * y_ref
    return m_t, c_t, m_w, c_w


class SpectrumTrace(object):
    """grism.py:479-669."""

    def __init__(self, x_ref, y_ref, trace_coeff, wl_solution):
        self.x_ref, self.y_ref = x_ref, y_ref
        self.m_t, self.c_t, self.m_w, self.c_w = wavelength_calibration_coeffs(x_ref, y_ref, trace_coeff,
                                                                               wl_solution)
        # _get_x_to_wl_poly_coeffs, grism.py:553-602
        x = np.array([x_ref + 10, x_ref + 20])
        y = self.x_to_y(x)
        d = np.sqrt((y - y_ref) ** 2 + (x - x_ref) ** 2)
        wl = (self.m_w * d + self.c_w) * 1e-4          # angstrom -> micron
        self.m_wl = (wl[1] - wl[0]) / (x[1] - x[0])
        self.c_wl = wl[0] - self.m_wl * x[0]

    def x_to_y(self, x):
        return self.m_t * (x - self.x_ref) + self.c_t + self.y_ref       # grism.py:537

    def wl_to_x(self, wl):
        return (wl - self.c_wl) / self.m_wl                              # grism.py:651

    def wl_to_y(self, wl):
        return self.x_to_y((wl - self.c_wl) / self.m_wl)                 # grism.py:667-669


class Grism(object):
    """G141 / G102 (grism.py:24-118, 426-476) over explicit calibration arrays:
    flat (4, 1014, 1014) float32, flat_wmin/max, sky (1014, 1014) float32,
    sens_wl (micron), sens_val."""

    def __init__(self, name, flat=None, flat_wmin=0., flat_wmax=1., sky=None, sens_wl=None, sens_val=None):
        self.name = name
        if name == "G141":
            self.trace_coeff, self.wl_solution, self.wl_limits = G141_TRACE, G141_WLSOL, (0.988, 1.777)
        elif name == "G102":
            self.trace_coeff, self.wl_solution, self.wl_limits = G102_TRACE, G102_WLSOL, (0.75, 1.2)
        else:
            raise ValueError(name)
        self.psf_ratio_poly = np.poly1d([-0.25063428, 0.8332488, -0.80546074, 0.39896516])   # grism.py:85-90
        self.psf_sigmal_poly = np.poly1d([0.69245668, -2.1043046, 2.22284446, -0.29689335])
        self.psf_sigmah_poly = np.poly1d([2.90366189, -8.81859432, 8.96049229, 2.254503])
        self.flat = flat
        self.flat_wmin, self.flat_wmax = flat_wmin, flat_wmax
        self.flat_xs, self.flat_ys = np.meshgrid(np.arange(1014), np.arange(1014))           # grism.py:69-70
        self.sky = sky
        self.throughput_wl, self.throughput_val = sens_wl, sens_val

    def set_current_wavelength_only_dependent_array(self, wl):
        """grism.py:111-118."""
        self.current_psf_ratio = self.psf_ratio_poly(wl)
        self.current_psf_sigmal = self.psf_sigmal_poly(wl)
        self.current_psf_sigmah = self.psf_sigmah_poly(wl)
        self.current_throughput_interpolated_function = np.interp(wl, self.throughput_wl, self.throughput_val)

    def _get_wavelength_calibration_coeffs(self, x_ref, y_ref):
        return wavelength_calibration_coeffs(x_ref, y_ref, self.trace_coeff, self.wl_solution)

    def get_trace(self, x_ref, y_ref):
        return SpectrumTrace(x_ref, y_ref, self.trace_coeff, self.wl_solution)

    def get_pixel_wl(self, x_ref, y_ref, x_1, y_1):
        """grism.py:137-163."""
        a_t, b_t, a_w, b_w = self._get_wavelength_calibration_coeffs(x_ref, y_ref)
        a_t_i = 1 / a_t
        d = np.sqrt((y_ref - y_1 + a_t_i * x_ref - a_t_i * x_1) ** 2 / (a_t_i ** 2 + 1))
        return a_w * d + b_w

    def get_pixel_wl_per_row(self, x_ref, y_ref, x_values=None, y_value=None):
        """grism.py:165-202."""
        x_values = np.arange(1014) if x_values is None else np.array(x_values)
        if y_value is None:
            y_value = y_ref
        a_t, b_t, a_w, b_w = self._get_wavelength_calibration_coeffs(x_ref, y_ref)
        a_t_i = 1 / a_t
        d_values = np.sqrt((y_ref - y_value + a_t_i * x_ref - a_t_i * x_values) ** 2 / (a_t_i ** 2 + 1))
        return a_w * d_values + b_w

    def _bin_centers_to_limits(self, centers, bin_size=1.):
        """grism.py:248-270."""
        centers = np.array(centers)
        half_bin = bin_size / 2.
        return np.append(centers - half_bin, centers[-1] + half_bin)

    def get_pixel_edges_wl_per_row(self, x_ref, y_ref, x_centers=None, y_value=None, pixel_size=1.):
        """grism.py:218-246."""
        return self.get_pixel_wl_per_row(x_ref, y_ref, self._bin_centers_to_limits(x_centers, pixel_size), y_value)

    def get_flat_field(self, x_ref, y_ref, size=None, indices=None, reference_quirks=False):
        """grism.py:349-409 (the `indices` branch the exposure path uses).

        size = 1024: the reference's (1014 - 1024) / 2 = -5 (py2 floor division) looks the flat up
        5 px up / left of the frame pixel -- the counterpart of its -5 frame offset
        (exposure_generator.py:630), negative indices wrapping as numpy's do -- and then fails on the
        empty crop_central_box (tools.py:322-324).  reference_quirks keeps the -5 (crop = identity);
        otherwise offset 0, the documented deviation for the 1014 frame."""
        f0, f1, f2, f3 = self.flat
        if size is not None:
            off = (1014 - size) // 2                              # py2 int division
            if size > 1014 and not reference_quirks:
                off = 0
            indices = (indices[0] + off, indices[1] + off)
        a_t, b_t, a_w, b_w = self._get_wavelength_calibration_coeffs(x_ref, y_ref)
        a_t_i = 1 / a_t
        arr_1 = y_ref - self.flat_ys[indices] + a_t_i * x_ref - a_t_i * self.flat_xs[indices]
        d_values = np.sqrt((arr_1 * arr_1) / (a_t_i * a_t_i + 1))
        wl_array = a_w * d_values + b_w
        wl_array_norm = (wl_array - self.flat_wmin) / (self.flat_wmax - self.flat_wmin)
        wl_array_norm_2 = wl_array_norm * wl_array_norm
        wl_array_norm_3 = wl_array_norm_2 * wl_array_norm
        flatfield = np.ones_like(f0)                               # float32, as the cube (:380)
        flatfield[indices] = (f0[indices] + (f1[indices] * wl_array_norm) + (f2[indices] * wl_array_norm_2) +
                              (f3[indices] * wl_array_norm_3))
        if size is not None:
            if size > 1014 and reference_quirks:
                # crop_central_box(flatfield, 1024) is EMPTY in the reference (index -5: array[-5:5]) and the
                # multiplication that follows raises; the crop it stands for -- cropped[y, x] =
                # flatfield[y + off, x + off], as for every sub-array -- is a shift by off = -5 here
                flatfield = np.roll(flatfield, (-off, -off), axis=(0, 1))
            else:
                flatfield = crop_central_box(flatfield, min(size, 1014))
        return flatfield

    def get_master_sky(self, size=None):
        """grism.py:411-423 (a fresh float32 array each call, as re-opening the file gives)."""
        sky_array = np.array(self.sky, dtype=np.float32)
        if size is not None:
            sky_array = np.array(crop_central_box(sky_array, size))
        return sky_array


# ---------------------------------------------------------------------------
# detector.py
# ---------------------------------------------------------------------------
class SampleModeError(Exception):
    pass


class Detector(object):
    """WFC3_IR (detector.py:16-350) over explicit calibration arrays: pfl
    (1014, 1014) float32 = gain file [5:-5, 5:-5]; lin (4, 1024, 1024) float32,
    UNCROPPED; dark_hdus = the HDU list of the mode's super-dark file as the
    reference opens it ([primary] + SCI, ERR, DQ, SAMP, TIME per read, last read
    first) -- this class picks the frames of a read itself (detector.py:183-190)."""

    def __init__(self, pfl=None, lin=None, dark_hdus=None, bias_256=None):
        self.min_counts, self.max_counts = -20, 78000             # detector.py:26-28
        self.constant_gain = 2.35                                 # detector.py:30
        self.read_noise = 14.1 / self.constant_gain               # detector.py:33
        with open(os.path.join(_DATA, "wfc3_ir_modes.json")) as f:
            t = json.load(f)
        self.modes_exp_table = {int(s): v for s, v in t["exptime"].items()}
        self.pfl, self.lin = pfl, lin
        self.dark_hdus = dark_hdus
        self.bias_256 = bias_256

    def _rows(self, NSAMP, SUBARRAY, SAMPSEQ):
        times = self.modes_exp_table.get(SUBARRAY, {}).get(SAMPSEQ, [])
        n = NSAMP - 1                                             # detector.py:86, 233
        if n < 1 or n > len(times):
            raise SampleModeError("SAMPSEQ = {}, NSAMP={}, SUBARRAY={} is not a permitted combination".format(
                SAMPSEQ, NSAMP, SUBARRAY))
        return times, n

    def exptime(self, NSAMP, SUBARRAY, SAMPSEQ):
        times, n = self._rows(NSAMP, SUBARRAY, SAMPSEQ)           # detector.py:69-100
        return times[n - 1]

    def get_read_times(self, NSAMP, SUBARRAY, SAMPSEQ):
        if not 2 <= NSAMP <= 16:                                  # detector.py:228-231
            raise SampleModeError("NSAMP must be an integer between 2 and 16, got {}".format(NSAMP))
        times, n = self._rows(NSAMP, SUBARRAY, SAMPSEQ)           # detector.py:233-246
        return np.array(times[:n])

    def gen_pixel_array(self, subarray, light_sensitive=True):
        """detector.py:102-124."""
        if light_sensitive:
            if subarray == 1024:
                subarray = 1014
            return np.zeros((subarray, subarray))
        size = subarray + 10
        if size > 1024:
            size = 1024
        return np.zeros((size, size))

    def add_bias_pixels(self, pixel_array):
        """detector.py:126-149."""
        array_size = len(pixel_array)
        if array_size not in (1014, 512, 256, 128, 64):
            raise ValueError("array size must be in (1014, 512, 256, 128, 64) got {}".format(array_size))
        full_array = np.zeros((array_size + 10, array_size + 10))
        full_array[5:-5, 5:-5] = pixel_array
        return full_array

    def get_gain(self, size):
        """detector.py:200-209: python float / float32 array stays float32."""
        gain_data = np.float32(self.constant_gain) / np.asarray(self.pfl, dtype=np.float32)
        if size is not None:
            gain_data = crop_central_box(gain_data, 1014 if size == 1024 else size)
        return gain_data

    def dark_for_read(self, read_NSAMP):
        """detector.py:183-190: file_index = -(NSAMP) * 5; f[file_index] is the dark frame,
        f[file_index + 1] its error; np.where(err > 0, err, 0.00001) keeps float32."""
        file_index = -(read_NSAMP) * 5
        dark_array = self.dark_hdus[file_index]
        err = self.dark_hdus[file_index + 1]
        return dark_array, np.where(err > 0, err, np.float32(0.00001))

    def apply_non_linearity(self, pixel_array):
        """detector.py:318-350: Newton-Raphson on the whole frame until EVERY
        pixel moved by < 1e-3 (so all pixels get the slowest pixel's iteration count)."""
        n = len(pixel_array)
        crop1 = len(self.lin[0]) // 2 - n // 2
        crop2 = len(self.lin[0]) // 2 + n // 2
        c1, c2, c3, c4 = (p[crop1:crop2, crop1:crop2] for p in self.lin)
        u0 = pixel_array
        u1 = u0 * 0
        for _ in range(10000):
            u1 = u0 - ((-pixel_array + u0 * (1 + c1 + u0 * (c2 + u0 * (c3 + c4 * u0)))) /
                       (1 + c1 + 2 * c2 * u0 + 3 * c3 * u0 * u0 + 4 * c4 * u0 * u0 * u0))
            if (np.abs(u1 - u0) < 10 ** (-3)).all():
                break
            u0 = u1
        return u1


# ---------------------------------------------------------------------------
# random draws
# ---------------------------------------------------------------------------
STAGE_COUNTS, STAGE_THROW, STAGE_SKY, STAGE_CR_COUNT, STAGE_CR_HIT, STAGE_READ, STAGE_NOISE, STAGE_HOST = \
    1, 2, 3, 4, 5, 6, 7, 8


class LegacyDraws(object):
    """numpy legacy generator, in the reference's call order."""
    philox = False

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)           # run_visit.py:68-77 seeds the global stream

    def sample_draws(self, K, x_jitter, y_jitter):
        seeds = self.rs.randint(0, 100000, K)           # exposure_generator.py:327-329
        return seeds, self.rs.normal(0, x_jitter, K), self.rs.normal(0, y_jitter, K)

    def stellar_poisson(self, lam, k):
        return self.rs.poisson(lam)                     # :626

    def gaussian_noise(self, mean, std, dim, r):
        return self.rs.normal(mean, std, (dim, dim))    # :725

    def sky_poisson(self, lam, r, unit_sky=None, bg_count=None):
        return self.rs.poisson(lam)                     # :495

    def cosmic_frame(self, rate, time, size, r):
        """cosmic_rays.py:88-139 (MinMaxPossionCosmicGenerator, 10000..35000)."""
        size_rate = rate / (1024. * 1024.) * (size * size)
        number = self.rs.poisson(size_rate * time)
        energies = self.rs.randint(10000, 35000, number)
        array = np.zeros((size, size))
        y_pos = self.rs.randint(0, size, number)
        x_pos = self.rs.randint(0, size, number)
        for i in range(number):
            array[y_pos[i], x_pos[i]] += energies[i]
        return array

    def dark_normal(self, dark, err, r):
        return self.rs.normal(dark, err)                # detector.py:191

    def read_normal(self, pixel_array, sigma, r):
        return self.rs.normal(pixel_array, sigma)       # detector.py:198


class PhiloxDraws(object):
    """The MI355X path's stream layout (wayne_amd/csrc/philox.h):
    key = (seed, stage), counter = (element, block, sub-sample | read, exposure);
    the per-pixel stages (SKY, READ, NOISE) and the thrower use one Philox block
    as the state of a short xoshiro128+ stream.  Pixel elements are indices
    into the BORDERED S x S frame."""
    philox = True

    def __init__(self, seed, exposure, N):
        self.seed, self.exposure, self.N, self.S = int(seed), int(exposure), N, N + 10
        yy, xx = np.mgrid[0:N, 0:N]
        self.interior_idx = np.ascontiguousarray(((yy + 5) * self.S + (xx + 5)).ravel().astype(np.uint32))
        self.all_idx = np.arange(self.S * self.S, dtype=np.uint32)
        self._pairs = []          # (z_dark, z_read) of read 0, 1, ... from the STAGE_READ streams
        self._state = {}
        self._sky_next = 0
        self._noise_next = 0

    def _blocks(self, c0, c1, c2, stage):
        c0 = np.ascontiguousarray(c0, dtype=np.uint32)
        out = np.empty((c0.size, 4), dtype=np.uint32)
        clib.lib().wayne_oracle_philox_blocks(c0, c0.size, c1, c2, self.exposure, self.seed, stage, out)
        return out

    def sample_draws(self, K, x_jitter, y_jitter):
        w = self._blocks(np.arange(K), 0, 0, STAGE_HOST)
        # x * 2^-32 + 2^-33 (a fused multiply-add on the host side; exact here: < 2^53)
        ua = w[:, 0].astype(np.float64) * 2.3283064365386963e-10 + 1.1641532182693481e-10
        ub = w[:, 1].astype(np.float64) * 2.3283064365386963e-10 + 1.1641532182693481e-10
        R = np.sqrt(-2.0 * np.log(ub))
        ang = 6.283185307179586476925 * ua
        seeds = ((w[:, 2].astype(np.uint64) * np.uint64(100000)) >> np.uint64(32)).astype(np.int64)
        return seeds, R * np.cos(ang) * x_jitter, R * np.sin(ang) * y_jitter

    def stellar_poisson(self, lam, k):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        out = np.empty(lam.size)
        clib.lib().wayne_oracle_poisson_counts_f64(lam, lam.size, self.seed, STAGE_COUNTS, 0, int(k), self.exposure, out)
        return out

    def _stream(self, name, idx, stage):
        """xoshiro128+ states of the seeded stream `stage` for the pixels `idx`."""
        if name not in self._state:
            st = np.empty((idx.size, 4), dtype=np.uint32)
            clib.lib().wayne_oracle_seed_streams(idx, idx.size, self.seed, stage, self.exposure, st)
            self._state[name] = st
        return self._state[name]

    def _normal_step(self, state):
        n = state.shape[0]
        z0 = np.empty(n, dtype=np.float32)
        z1 = np.empty(n, dtype=np.float32)
        clib.lib().wayne_oracle_normal_step(n, state, z0, z1)
        return z0, z1

    def gaussian_noise(self, mean, std, dim, r):
        # STAGE_NOISE stream of each interior pixel: words 2r, 2r+1 -> read interval r (calls come in order)
        assert r == self._noise_next
        self._noise_next += 1
        z0, _ = self._normal_step(self._stream("noise", self.interior_idx, STAGE_NOISE))
        return mean + std * z0.astype(np.float64).reshape(dim, dim)

    SKY_TABLE = 256          # entries per alias table
    SKY_TABLES = 15          # tables that fit the device's LDS array (kMaxReads)

    @staticmethod
    def _sky_fits(lam):
        lam = float(lam)
        return lam >= 0. and lam + 8. * np.sqrt(lam) + 8. <= 255.

    def begin_sky(self, unit_sky, bg_counts):
        """Plan the sky draws of an exposure (device: wayne_exposure_upload + k_ramp.h sky_draw):
        L levels of the master sky per distinct read interval, one alias table of
        Poisson(level * bg_count) each; used when every table fits 256 entries."""
        unit = np.asarray(unit_sky, dtype=np.float32)
        bg_counts = [np.float32(b) for b in bg_counts]
        self._sky_plan = None
        pos = unit[unit > 0]
        if pos.size == 0:
            return
        distinct = []
        for b in bg_counts:
            if not any(b.tobytes() == d.tobytes() for d in distinct):
                distinct.append(b)
        L = max(1, min(self.SKY_TABLES // len(distinct), self.SKY_TABLES))
        ordered = np.sort(pos)
        levels = np.array([ordered[(l * ordered.size) // L] for l in range(L)], dtype=np.float32)   # l/L quantiles
        tables = np.zeros((len(distinct) * L, self.SKY_TABLE), dtype=np.uint32)
        for j, b in enumerate(distinct):
            for l in range(L):
                lam = np.float32(levels[l] * b)
                if not self._sky_fits(lam):
                    return                                    # some read does not fit: direct sampler
                clib.lib().wayne_oracle_sky_alias_table(float(lam), tables[j * L + l])
        # the highest level not above the pixel (level 0, the minimum, for the pixels that draw nothing anyway)
        lvl = np.zeros(unit.shape, dtype=np.int64)
        for l in range(1, L):
            lvl += (levels[l] <= unit)
        self._sky_plan = dict(L=L, distinct=distinct, levels=levels, tables=tables, lvl=lvl)

    def sky_poisson(self, lam, r, unit_sky=None, bg_count=None):
        # STAGE_SKY stream of each interior pixel, consumed read after read
        assert r == self._sky_next
        self._sky_next += 1
        lam32 = np.ascontiguousarray(lam, dtype=np.float32).ravel()
        out = np.empty(lam32.size)
        plan = getattr(self, "_sky_plan", None)
        if plan is None:
            # the direct sampler (a data-dependent number of words per draw) has a stream of its own
            state = self._stream("sky", self.interior_idx, STAGE_SKY)
            clib.lib().wayne_oracle_poisson_sky_step(lam32, lam32.size, state, out)
            return out.reshape(lam.shape)
        # the table-driven draw reads the pixel's STAGE_READ stream, between the normals of read r (0 = zero read) and
        # those of read r + 1 (philox.h): bring the stream to that point, then step the interior pixels' states
        self._read_pair(r)
        st_all = self._stream("read", self.all_idx, STAGE_READ)
        state = np.ascontiguousarray(st_all[self.interior_idx])
        j = [i for i, d in enumerate(plan["distinct"]) if d.tobytes() == np.float32(bg_count).tobytes()][0]
        lvl = plan["lvl"].ravel()
        lam_level = np.ascontiguousarray((plan["levels"][lvl] * np.float32(bg_count)).astype(np.float32))
        table_of = np.ascontiguousarray((j * plan["L"] + lvl).astype(np.int32))
        clib.lib().wayne_oracle_sky_alias_step(lam32, lam_level, table_of, plan["tables"].ravel(), lam32.size, state, out)
        st_all[self.interior_idx] = state
        return out.reshape(lam.shape)

    def cosmic_frame(self, rate, time, size, r):
        size_rate = rate / (1024. * 1024.) * float(size * size)
        lam = np.array([size_rate * time])
        n = np.empty(1)
        clib.lib().wayne_oracle_poisson_f64(lam, 1, self.seed, STAGE_CR_COUNT, 0, int(r), self.exposure, n)
        number = int(min(max(n[0], 0), 1e7))
        array = np.zeros((size, size))
        if number:
            w = self._blocks(np.arange(number), 0, int(r), STAGE_CR_HIT).astype(np.uint64)
            energies = 10000 + ((w[:, 0] * np.uint64(25000)) >> np.uint64(32)).astype(np.int64)
            y_pos = ((w[:, 1] * np.uint64(size)) >> np.uint64(32)).astype(np.int64)
            x_pos = ((w[:, 2] * np.uint64(size)) >> np.uint64(32)).astype(np.int64)
            np.add.at(array, (y_pos, x_pos), energies)
        return array

    def _read_pair(self, r):
        # the pair of normals of read i (0 = zero read) from every pixel's STAGE_READ stream; the table-driven sky
        # draw of read interval i takes its words from the same stream right after them (sky_poisson)
        st = self._stream("read", self.all_idx, STAGE_READ)
        while len(self._pairs) <= r:
            self._pairs.append(self._normal_step(st))
        return self._pairs[r]

    def dark_normal(self, dark, err, r):
        z0, _ = self._read_pair(r)
        return dark + err.astype(np.float64) * z0.astype(np.float64).reshape(dark.shape)

    def read_normal(self, pixel_array, sigma, r):
        _, z1 = self._read_pair(r)
        return pixel_array + sigma * z1.astype(np.float64).reshape(pixel_array.shape)


# ---------------------------------------------------------------------------
# exposure_generator.py
# ---------------------------------------------------------------------------
class ExposureOracle(object):
    """ExposureGenerator (exposure_generator.py:16-727), reads returned as a
    list of NSAMP float64 (S, S) arrays, read 0 = zero read."""

    def __init__(self, detector, grism, NSAMP, SAMPSEQ, SUBARRAY):
        self.detector, self.grism = detector, grism
        self.NSAMP, self.SAMPSEQ, self.SUBARRAY = NSAMP, SAMPSEQ, SUBARRAY
        self.exptime = detector.exptime(NSAMP, SUBARRAY, SAMPSEQ)                 # :54
        self.read_times = detector.get_read_times(NSAMP, SUBARRAY, SAMPSEQ)       # :57 (seconds)

    def _gen_scanning_sample_times(self, sample_rate):
        """exposure_generator.py:531-579 (milliseconds)."""
        read_times = self.read_times * 1000.
        read_index = []
        i = -1
        sample_starts = []
        previous_read = 0.
        for read_time in read_times:
            starts = np.arange(previous_read, read_time, sample_rate)
            sample_starts.append(starts)
            i += len(starts)
            read_index.append(i)
            previous_read = read_time
        sample_starts = np.concatenate(sample_starts)
        _ends = np.roll(sample_starts, -1)
        _ends[-1] = read_times[-1]
        sample_durations = _ends - sample_starts
        sample_mid_points = sample_starts + (sample_durations / 2)
        return sample_starts, sample_mid_points, sample_durations, read_index

    def _gen_sample_yref(self, y_ref, mid_points, scan_speed):
        return y_ref + (mid_points * scan_speed)                                  # :527

    def _gen_zero_read(self, add_initial_bias=True):
        """exposure_generator.py:446-466."""
        pixel_array_full = self.detector.gen_pixel_array(self.SUBARRAY, light_sensitive=False)
        if self.SUBARRAY == 256 and add_initial_bias:
            pixel_array_full += self.detector.bias_256
        return pixel_array_full

    def counts_before_noise(self, wl, flux, exptime, scale_factor):
        """The counts chain of _gen_subsample / _flux_to_counts (:602-623, :649-687):
        flux * sensitivity [e/(s A)] * delta_lambda [um] * 1e4 [A/um] * exptime [ms] * 1e-3 [s/ms] * scale."""
        count_rate = flux * self.grism.current_throughput_interpolated_function
        delta_lambda = bin_centers_to_widths(wl)
        count_rate = count_rate * delta_lambda
        count_rate = count_rate * 1e4
        counts = count_rate * exptime
        counts = counts * 1e-3
        if scale_factor is not None:
            counts = counts * scale_factor
        return counts

    def _gen_subsample(self, x_ref, y_ref, wl, flux, pixel_array, exptime, rand_seed, threads, scale_factor,
                       add_flat, add_stellar_noise, draws, k, thrower, sub_scale):
        """exposure_generator.py:581-647."""
        trace = self.grism.get_trace(x_ref, y_ref)
        x_pos = trace.wl_to_x(wl)
        y_pos = trace.wl_to_y(wl)
        psf_ratio = self.grism.current_psf_ratio
        psf_sigmal = self.grism.current_psf_sigmal
        psf_sigmah = self.grism.current_psf_sigmah
        counts = self.counts_before_noise(wl, flux, exptime, scale_factor)
        if add_stellar_noise:
            counts = draws.stellar_poisson(counts, k)        # :626
        else:
            counts = np.round(counts)                        # :628
        x_sub = x_pos - sub_scale                            # :630-632
        y_sub = y_pos - sub_scale
        y_size, x_size = pixel_array.shape
        frame = thrower(counts, x_sub, y_sub, psf_ratio, psf_sigmal, psf_sigmah, y_size, x_size, rand_seed,
                        threads, k)
        new_pixel_array = np.reshape(frame, (y_size, x_size)).astype(np.float64)   # pyparallel.pyx:31-34
        if add_flat:
            flat_field = self.grism.get_flat_field(x_ref, y_ref, self.SUBARRAY, np.where(new_pixel_array > 0),
                                                   reference_quirks=getattr(self, "_quirks", False))
            new_pixel_array *= flat_field                    # :641-645
        return new_pixel_array, counts, x_sub, y_sub

    def _add_read_reductions(self, pixel_array, read_exp_time, noise_mean, noise_std, sky_background,
                             add_gain_variations, cosmic_rate, draws, r):
        """exposure_generator.py:468-515."""
        array_size = pixel_array.shape[0]
        if noise_mean and noise_std:
            pixel_array += draws.gaussian_noise(noise_mean * read_exp_time, noise_std * read_exp_time,
                                                array_size, r)
        if sky_background:
            master_sky = self.grism.get_master_sky(array_size)
            bg_count = sky_background * read_exp_time
            unit_sky = master_sky.copy()
            master_sky *= np.float32(bg_count)               # in-place on a float32 array (:493)
            pixel_array += draws.sky_poisson(master_sky, r, unit_sky=unit_sky, bg_count=np.float32(bg_count))
        if cosmic_rate is not None:
            pixel_array += draws.cosmic_frame(cosmic_rate, read_exp_time, array_size, r)
        if add_gain_variations:
            pixel_array /= self.detector.get_gain(self.SUBARRAY)
        else:
            pixel_array /= self.detector.constant_gain
        return self.detector.add_bias_pixels(pixel_array)

    def _post_exposure_reductions(self, reads, add_dark, add_non_linear, clip_values_det_limits, add_read_noise,
                                  draws):
        """exposure_generator.py:407-444 with the Exposure methods it calls (exposure.py:49-131)."""
        if add_dark and self.detector.dark_hdus is not None:
            for i in range(1, len(reads)):                   # exposure.py:70-80
                dark, err = self.detector.dark_for_read(i + 1)
                reads[i] = reads[i] + draws.dark_normal(dark, err, i)
        if add_non_linear:
            for i in range(1, len(reads)):                   # exposure.py:49-59
                reads[i] = self.detector.apply_non_linearity(reads[i])
        if clip_values_det_limits:
            for i in range(len(reads)):                      # exposure.py:82-92
                reads[i] = np.clip(reads[i], self.detector.min_counts, self.detector.max_counts)
        for i in range(len(reads)):                          # exposure.py:122-131
            ref_is_true = np.ones_like(reads[i], dtype="bool_")
            ref_is_true[5:-5, 5:-5] = 0
            reads[i][ref_is_true] = 0.
        zero_read = reads[0]
        for i in range(1, len(reads)):                       # exposure.py:94-104
            reads[i] = reads[i] + zero_read
        if add_read_noise:
            for i in range(len(reads)):                      # exposure.py:61-68
                reads[i] = draws.read_normal(reads[i], self.detector.read_noise, i)
        return reads

    def scanning_frame(self, x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal, scan_speed,
                       sample_rate, sample_mid_points=None, sample_durations=None, read_index=None,
                       ssv_generator=None, noise_mean=False, noise_std=False, add_dark=True, add_flat=True,
                       cosmic_rate=None, sky_background=1.0, scale_factor=None, add_gain_variations=True,
                       add_non_linear=True, clip_values_det_limits=True, add_read_noise=True,
                       add_stellar_noise=True, add_initial_bias=True, threads=2, draws=None,
                       thrower="oracle", reference_quirks=False, record=None):
        """exposure_generator.py:178-405.  scan_speed px/s, sample_rate ms.

        `thrower`: 'ref' = the reference's compiled C (oracle/_ref), 'oracle' =
        the C restatement (both use rand_seed / threads), 'philox' = the
        Philox-keyed thrower.  `record`, if a dict, receives intermediates."""
        scan_speed = scan_speed / 1000.                      # px/ms (:247)
        self._quirks = bool(reference_quirks)
        if sample_mid_points is None and sample_durations is None and read_index is None:
            _, sample_mid_points, sample_durations, read_index = self._gen_scanning_sample_times(sample_rate)
        s_y_refs = self._gen_sample_yref(y_ref, sample_mid_points, scan_speed)    # :258
        if ssv_generator is not None:
            if isinstance(ssv_generator, SSVModulatedSine):                        # :263-267
                sample_durations, read_index = ssv_generator.get_subsample_exposure_times(
                    s_y_refs, sample_durations, self.read_times, sample_rate)
            else:
                sample_durations = ssv_generator.get_subsample_exposure_times(
                    s_y_refs, sample_durations, self.read_times, sample_rate)     # :272-273

        zero_read = self._gen_zero_read(add_initial_bias)                         # :302
        reads = [zero_read.copy()]
        cumulative_pixel_array = self.detector.gen_pixel_array(self.SUBARRAY, light_sensitive=False)
        read_num = 0
        read_exp_times = self.read_times
        previous_read_time = 0.
        pixel_array = self.detector.gen_pixel_array(self.SUBARRAY, light_sensitive=True)
        num_samples = len(sample_mid_points)
        s_rand_seeds, s_x_jitter, s_y_jitter = draws.sample_draws(num_samples, x_jitter, y_jitter)   # :327-329
        crop_ind = crop_spectrum_ind(self.grism.wl_limits[0], self.grism.wl_limits[-1], wl.copy())   # :332
        s_wl = wl[crop_ind[0]:crop_ind[1]]
        # :630 -- py2: 507 - SUBARRAY/2; -5 at 1024, which the build replaces by 0
        sub_scale = 507 - (self.SUBARRAY // 2)
        if self.SUBARRAY == 1024 and not reference_quirks:
            sub_scale = 0

        if thrower == "ref":
            def throw(counts, x, y, ratio, sl, sh, ny, nx, seed, threads, k):
                return clib.psf_reference(np.asarray(counts).astype(np.int32), x, y, ratio, sl, sh, ny, nx,
                                          int(seed), threads)
        elif thrower == "oracle":
            def throw(counts, x, y, ratio, sl, sh, ny, nx, seed, threads, k):
                return clib.psf_oracle(np.asarray(counts).astype(np.int32), x, y, ratio, sl, sh, ny, nx,
                                       int(seed), threads)
        elif thrower == "philox":
            def throw(counts, x, y, ratio, sl, sh, ny, nx, seed, threads, k):
                return clib.psf_philox_oracle(np.asarray(counts).astype(np.int32), x, y, ratio, sl, sh, ny, nx,
                                              draws.seed, draws.exposure, k)
        elif thrower == "split":
            def throw(counts, x, y, ratio, sl, sh, ny, nx, seed, threads, k):
                return clib.psf_split_oracle(np.asarray(counts).astype(np.int32), x, y, ratio, sl, sh, ny,
                                             draws.seed, draws.exposure, k)
        else:
            raise ValueError(thrower)

        if record is not None:
            record.update(counts=[], x=[], y=[], acc=[])
        if sky_background and hasattr(draws, "begin_sky"):
            dts = np.diff(np.concatenate([[0.], np.asarray(read_exp_times, dtype=float)]))
            draws.begin_sky(self.grism.get_master_sky(pixel_array.shape[0]), [sky_background * dt for dt in dts])
        wavelength_only_test = False
        for i, s_mid in enumerate(sample_mid_points):                             # :336
            try:
                s_y_ref = s_y_refs[i]
                s_dur = sample_durations[i]
            except IndexError:                                                    # :340-342
                s_dur = 0.
                s_y_ref = s_y_refs[-1]
            if planet_signal is not None:                                         # :344-348
                s_flux = stellar_flux[crop_ind[0]:crop_ind[1]] * (1. - planet_signal[i][crop_ind[0]:crop_ind[1]])
            else:
                s_flux = stellar_flux[crop_ind[0]:crop_ind[1]]
            if not wavelength_only_test:                                          # :350-352
                self.grism.set_current_wavelength_only_dependent_array(s_wl)
                wavelength_only_test = True
            sample_frame, counts, xs, ys = self._gen_subsample(
                x_ref + s_x_jitter[i], s_y_ref + s_y_jitter[i], s_wl, s_flux, pixel_array, s_dur,
                s_rand_seeds[i], threads, scale_factor, add_flat, add_stellar_noise, draws, i, throw, sub_scale)
            pixel_array += sample_frame                                           # :359
            if record is not None:
                record["counts"].append(np.asarray(counts).astype(np.int64))
                record["x"].append(xs)
                record["y"].append(ys)
            if i in read_index:                                                   # :361
                read_exp_time = read_exp_times[read_num] - previous_read_time
                if record is not None:
                    record["acc"].append(self.detector.add_bias_pixels(pixel_array.copy()))
                pixel_array_full = self._add_read_reductions(
                    pixel_array, read_exp_time, noise_mean, noise_std, sky_background, add_gain_variations,
                    cosmic_rate, draws, read_num)
                cumulative_pixel_array += pixel_array_full                        # :378
                reads.append(cumulative_pixel_array.copy())                       # :381
                previous_read_time = read_exp_times[read_num]
                read_num += 1
                pixel_array = self.detector.gen_pixel_array(self.SUBARRAY, light_sensitive=True)   # :388
        assert len(reads) == self.NSAMP                                           # :397
        return self._post_exposure_reductions(reads, add_dark, add_non_linear, clip_values_det_limits,
                                              add_read_noise, draws)

    def staring_frame(self, x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal,
                      sample_mid_points, sample_durations, read_index, **kw):
        """exposure_generator.py:146-176: scan speed 0, one sample per read."""
        return self.scanning_frame(x_ref, y_ref, x_jitter, y_jitter, wl, stellar_flux, planet_signal, 0.,
                                   365.25 * 86400. * 1000., sample_mid_points, sample_durations, read_index,
                                   None, **kw)


class SSVSine(object):
    """scan_speed_varations.py:13-60 with a numeric start phase."""

    def __init__(self, stddev=1.5, period=0.7, start_phase=0.):
        self.stddev, self.period, self.start_phase = stddev, period, start_phase

    def get_subsample_exposure_times(self, y_mid_points, sample_durations, subsample_exptime=None,
                                     total_exptime=None):
        zeroed_y_mid = y_mid_points - y_mid_points[0]
        ssv_scaling = (self.stddev / 100.) * np.sin((self.period * zeroed_y_mid) + self.start_phase) + 1.
        return sample_durations * ssv_scaling


class SSVModulatedSine(object):
    """scan_speed_varations.py:63-171, statement for statement and draw for draw, over a numpy legacy generator
    `rs` standing in for the reference's global np.random stream (run_visit.py:68-77), with the astropy units
    written out: `read_times` in seconds, `sample_rate` in MILLISECONDS (what ExposureGenerator passes,
    exposure_generator.py:263-267; the reference converts both to seconds, :90-91).  Returns (durations in ms,
    read indexes) as the reference does (:171)."""

    def __init__(self, amplitude=10, period=1.1, blip_proba=1, rs=None):
        self.amplitude, self.period, self.blip_proba = amplitude, period, blip_proba      # :78-80
        self.rs = rs     # the stream a call draws from when none is handed to it

    def get_subsample_exposure_times(self, y_mid_points, sample_durations, read_times, sample_rate, rs=None):
        rs = self.rs if rs is None else rs
        read_times = np.asarray(read_times, dtype=float)             # :90
        sample_rate = float(sample_rate) / 1000.                     # :91 (ms -> s)
        period = self.period
        amplitude = self.amplitude
        exptime = np.round(read_times[-1], 6)                        # :96
        tt = np.arange(0, exptime, sample_rate)                      # :97
        amp1 = np.ones_like(tt)
        amp2 = rs.normal(0.1, 0.05) * np.sin(
            (2 * np.pi / rs.normal(2.0 * exptime, 0.5 * exptime)) * tt + rs.random_sample() * 2 * np.pi)   # :100-102
        if 100.0 * rs.random_sample() < self.blip_proba:             # :103
            amp3 = rs.normal(1.0, 0.1) * np.exp(-(tt - rs.random_sample() * exptime) ** 2 / (2 * (period / 2) ** 2))
        else:
            amp3 = 0
        final_amp = sample_rate * (amplitude / 100.0) * (amp1 + amp2 + amp3)     # :110
        per1 = np.ones_like(tt)
        per2 = rs.normal(0.1, 0.05) * np.sin(
            (2 * np.pi / rs.normal(2.0 * exptime, 0.5 * exptime)) * tt + rs.random_sample() * 2 * np.pi)   # :113-115
        final_per = period * (per1 + per2)
        final_phase = rs.random_sample() * 2 * np.pi                 # :118
        final_sub_exptimes = np.round(sample_rate + final_amp * np.sin((2 * np.pi / final_per) * tt + final_phase), 6)
        difference = int((10 ** 6) * np.round(exptime - np.sum(final_sub_exptimes), 6))   # :122-123
        if difference < 0:
            for i in range(abs(difference)):
                final_sub_exptimes[rs.randint(len(final_sub_exptimes))] -= 0.000001
        else:
            for i in range(abs(difference)):
                final_sub_exptimes[rs.randint(len(final_sub_exptimes))] += 0.000001
        breaks = []
        for i in read_times:                                         # :133-135
            breaks.append(int(np.argmin(abs(np.cumsum(final_sub_exptimes) - i))))
        difference = int((10 ** 6) * np.round(read_times[0] - np.sum(final_sub_exptimes[:breaks[0] + 1]), 6))
        dis = np.int_(rs.power(3, abs(difference)) * (breaks[0] + 1))             # :139
        if difference < 0:
            for i in dis:
                final_sub_exptimes[breaks[0] - i] -= 0.000001
                final_sub_exptimes[rs.randint(breaks[0] + 1, len(final_sub_exptimes))] += 0.000001
        else:
            for i in dis:
                final_sub_exptimes[breaks[0] - i] += 0.000001
                final_sub_exptimes[rs.randint(breaks[0] + 1, len(final_sub_exptimes))] -= 0.000001
        for read in range(1, len(read_times) - 1):                   # :151-167
            difference = int((10 ** 6) * np.round(read_times[read] - np.sum(final_sub_exptimes[:breaks[read] + 1]), 6))
            if difference < 0:
                for i in range(abs(difference)):
                    final_sub_exptimes[rs.randint(breaks[read - 1] + 1, breaks[read] + 1)] -= 0.000001
                    final_sub_exptimes[rs.randint(breaks[read] + 1, len(final_sub_exptimes))] += 0.000001
            else:
                for i in range(abs(difference)):
                    final_sub_exptimes[rs.randint(breaks[read - 1] + 1, breaks[read] + 1)] += 0.000001
                    final_sub_exptimes[rs.randint(breaks[read] + 1, len(final_sub_exptimes))] -= 0.000001
        read_indexes = breaks
        return final_sub_exptimes * 1000., read_indexes              # :171 (s -> ms)


def from_calibration(cal, grism_name, NSAMP, SAMPSEQ, SUBARRAY, g102_flat_quirk=False):
    """Build (Detector, Grism, ExposureOracle) over the UNCROPPED arrays of a
    calibration set (any object with .flat/.flat_wl/.sky/.sens/.pfl/.lin/.bias_256
    and .super_dark_hdus(), e.g. wayne_amd.calibration.CalibrationSet -- data
    access only: every crop and every choice of a read's dark frame is made here,
    by the restatement, not by a product helper)."""
    det = Detector(pfl=cal.pfl, lin=cal.lin, bias_256=cal.bias_256)
    try:
        det.dark_hdus = cal.super_dark_hdus(SUBARRAY, SAMPSEQ)
    except BaseException as e:  # no super-dark for the mode
        if type(e).__name__ != "WFC3SimNoDarkFileError":
            raise
    # the reference's G102 keeps the flat cube (and WMIN / WMAX) that G141.__init__ loaded (grism.py:428,
    # :66-76; only the path attribute is overridden, :453-454): reproduced when g102_flat_quirk is set
    flat_name = "G141" if (g102_flat_quirk and grism_name == "G102") else grism_name
    wmin, wmax = cal.flat_wl.get(flat_name, (0., 1.))
    sw, sv = cal.sens[grism_name]
    gr = Grism(grism_name, flat=cal.flat.get(flat_name), flat_wmin=wmin, flat_wmax=wmax,
               sky=cal.sky.get(grism_name), sens_wl=sw, sens_val=sv)
    return det, gr, ExposureOracle(det, gr, NSAMP, SAMPSEQ, SUBARRAY)
