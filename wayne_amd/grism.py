"""WFC3-IR grisms G141 and G102: field-dependent trace and dispersion
solutions, PSF polynomials, sensitivity, flat cube geometry.

Host-side mirror of the reference's wayne/grism.py (constants :56-118,
:426-476, :756-776; trace class :479-687; aXe polynomials :779-803).  On the
exposure path these formulas run on the GPU (k_prep_wl / k_prep_sub /
flat_value in wayne_amd/csrc/kernels.h); the numpy versions here serve the
caller-facing API (pixel wavelengths, trace inspection) and feed the device
descriptor.  Plotting helpers of the reference are not provided.
"""
import numpy as np

# aXe configuration polynomials (Kuntschner et al. 2009, WFC3 ISRs 2009-17/18),
# as tabulated in grism.py:756-776.
g141_trace_coeff = (1.96882, 9.09159E-5, -1.93260E-3, 1.04275E-2, -7.96978E-6,
                    -2.49607E-6, 1.45963E-9, 1.39757E-8, 4.8494E-10)
g102_trace_coeff = (-3.55018E-1, 3.28722E-5, -1.44571E-3, 1.42852E-2,
                    -7.20713E-6, -2.42542E-6, 1.18294E-9, 1.19634E-8, 6.17274E-10)
g141_wl_solution = (8.95431E3, 9.35925E-2, 0, 4.51423E1, 3.17239E-4,
                    2.17055E-3, -7.42504E-7, 3.48639E-7, 3.09213E-7)
g102_wl_solution = (6.38738E3, 4.55507E-2, 0, 2.35716E1, 3.60396E-4,
                    1.58739E-3, -4.25234E-7, -6.53726E-8, 0.)


def wavelength_calibration_coeffs(x_ref, y_ref, trace_coeff, wl_sol_coeff):
    """(m_t, c_t, m_w, c_w): trace slope / offset and dispersion (A/px) /
    zero point (A) at the source position (grism.py:779-803)."""
    a, b = trace_coeff, wl_sol_coeff
    m_t = a[3] + a[4] * x_ref + a[5] * y_ref + a[6] * x_ref ** 2 + a[7] * x_ref * y_ref + a[8] * y_ref ** 2
    c_t = a[0] + a[1] * x_ref + a[2] * y_ref
    m_w = b[3] + b[4] * x_ref + b[5] * y_ref + b[6] * x_ref ** 2 + b[7] * x_ref * y_ref + b[8] * y_ref ** 2
    c_w = (b[0] + b[1] * x_ref) + b[2] * y_ref
    return m_t, c_t, m_w, c_w


class _SpectrumTrace(object):
    """Trace line and wavelength <-> position map of one source (grism.py:479-687)."""

    def __init__(self, x_ref, y_ref, trace_coeff, wl_solution):
        self.x_ref, self.y_ref = x_ref, y_ref
        self.trace_coeff, self.wl_solution = trace_coeff, wl_solution
        self.m_t, self.c_t, self.m_w, self.c_w = self._get_wavelength_calibration_coeffs(x_ref, y_ref)
        self.m_wl, self.c_wl = self._get_x_to_wl_poly_coeffs(x_ref, y_ref)

    def _get_wavelength_calibration_coeffs(self, x_ref, y_ref):
        return wavelength_calibration_coeffs(x_ref, y_ref, self.trace_coeff, self.wl_solution)

    def x_to_y(self, x):
        return self.m_t * (x - self.x_ref) + self.c_t + self.y_ref

    def y_to_x(self, y):
        return ((y - self.y_ref - self.c_t) / self.m_t) + self.x_ref

    def _get_x_to_wl_poly_coeffs(self, x_ref, y_ref):
        # straight line through the wavelengths (micron) of the trace points
        # at x_ref + 10 and x_ref + 20 (grism.py:553-602)
        x = np.array([x_ref + 10, x_ref + 20])
        y = self.x_to_y(x)
        d = np.sqrt((y - y_ref) ** 2 + (x - x_ref) ** 2)
        wl = (self.m_w * d + self.c_w) * 1e-4
        m_wl = (wl[1] - wl[0]) / (x[1] - x[0])
        return m_wl, wl[0] - m_wl * x[0]

    def x_to_wl(self, x):
        return self.m_wl * x + self.c_wl

    def y_to_wl(self, y):
        return self.x_to_wl(self.y_to_x(y))

    def wl_to_x(self, wl):
        return (np.asarray(wl, dtype=float) - self.c_wl) / self.m_wl

    def wl_to_y(self, wl):
        return self.x_to_y(self.wl_to_x(wl))

    def psf_line(self, wl):
        x, y = self.wl_to_x(wl), self.wl_to_y(wl)
        m = -np.array(1.) / self.m_t
        return x, y, m, y - m * x

    def xangle(self):
        return np.arctan(self.m_t)

    def psf_length_per_pixel(self):
        return 1 / np.cos(self.xangle())


class G141_Trace(_SpectrumTrace):
    def __init__(self, x_ref, y_ref):
        _SpectrumTrace.__init__(self, x_ref, y_ref, g141_trace_coeff, g141_wl_solution)
        self.grism_name = "G141"


class G102_Trace(_SpectrumTrace):
    def __init__(self, x_ref, y_ref):
        _SpectrumTrace.__init__(self, x_ref, y_ref, g102_trace_coeff, g102_wl_solution)
        self.grism_name = "G102"


class G141(object):
    """G141 grism.  `calibration` is a wayne_amd.calibration.CalibrationSet
    (sensitivity table, flat cube, master sky); the reference opens the same
    data from params._calb_dir at construction (grism.py:66-106)."""

    name = "G141"
    trace = G141_Trace
    trace_coeff = g141_trace_coeff
    wl_solution = g141_wl_solution
    min_lambda, max_lambda = 1.075, 1.7       # micron (grism.py:58-59)
    resolution = 130
    wl_limits = (0.988, 1.777)                # crop limits, micron (grism.py:94)
    # double-gaussian PSF polynomials in wavelength [micron] (grism.py:85-90)
    psf_ratio_poly = np.poly1d([-0.25063428, 0.8332488, -0.80546074, 0.39896516])
    psf_sigmal_poly = np.poly1d([0.69245668, -2.1043046, 2.22284446, -0.29689335])
    psf_sigmah_poly = np.poly1d([2.90366189, -8.81859432, 8.96049229, 2.254503])

    def __init__(self, calibration=None):
        self.calibration = calibration

    # -- trace / wavelength solution -----------------------------------------
    def _get_wavelength_calibration_coeffs(self, x_ref, y_ref):
        return wavelength_calibration_coeffs(x_ref, y_ref, self.trace_coeff, self.wl_solution)

    def get_trace(self, x_ref, y_ref):
        return self.trace(x_ref, y_ref)

    def get_pixel_wl(self, x_ref, y_ref, x_1, y_1):
        """Wavelength (A) of pixel (x_1, y_1) for a source at (x_ref, y_ref) (grism.py:137-163)."""
        a_t, _, a_w, b_w = self._get_wavelength_calibration_coeffs(x_ref, y_ref)
        a_t_i = 1 / a_t
        d = np.sqrt((y_ref - y_1 + a_t_i * x_ref - a_t_i * x_1) ** 2 / (a_t_i ** 2 + 1))
        return a_w * d + b_w

    def get_pixel_wl_per_row(self, x_ref, y_ref, x_values=None, y_value=None):
        x_values = np.arange(1014) if x_values is None else np.array(x_values)
        if y_value is None:
            y_value = y_ref
        return self.get_pixel_wl(x_ref, y_ref, x_values, y_value)

    def get_pixel_wl_whole_detector(self, x_ref, y_ref):
        ys, xs = np.mgrid[0:1014, 0:1014]
        return self.get_pixel_wl(x_ref, y_ref, xs, ys)

    def _bin_centers_to_limits(self, centers, bin_size=1.):
        centers = np.array(centers)
        half = bin_size / 2.
        return np.append(centers - half, centers[-1] + half)

    def get_pixel_edges_wl_per_row(self, x_ref, y_ref, x_centers=None, y_value=None, pixel_size=1.):
        return self.get_pixel_wl_per_row(x_ref, y_ref, self._bin_centers_to_limits(x_centers, pixel_size), y_value)

    # -- wavelength-only arrays ------------------------------------------------
    def set_current_wavelength_only_dependent_array(self, wl):
        """PSF parameters and sensitivity on the grid `wl` (grism.py:111-118)."""
        self.current_psf_ratio = self.psf_ratio_poly(wl)
        self.current_psf_sigmal = self.psf_sigmal_poly(wl)
        self.current_psf_sigmah = self.psf_sigmah_poly(wl)
        sens_wl, sens_val = self.calibration.sensitivity(self.name)
        self.current_throughput_interpolated_function = np.interp(wl, sens_wl, sens_val)

    def apply_throughput(self, wl, flux):
        sens_wl, sens_val = self.calibration.sensitivity(self.name)
        return flux * np.interp(wl, sens_wl, sens_val)


class G102(G141):
    name = "G102"
    trace = G102_Trace
    trace_coeff = g102_trace_coeff
    wl_solution = g102_wl_solution
    min_lambda, max_lambda = 0.8, 1.15        # grism.py:442-443
    resolution = 210
    wl_limits = (0.75, 1.2)                   # grism.py:464
