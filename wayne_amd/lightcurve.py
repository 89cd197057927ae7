"""Transit / eclipse light curves for the visit driver.

The reference evaluates, per exposure, one `pylightcurve.transit('claret', ...)`
and one `pylightcurve.eclipse(...)` call PER WAVELENGTH ELEMENT (W ~ 4.5 k
Python-level calls, observation.py:349-355).  pylightcurve is not available
here, so this module states the model itself (parity unpinned against
pylightcurve; validated against brute-force integration and the analytic
uniform-disk formula, tests/test_lightcurve.py):

  * orbit: Keplerian, mid-transit time T0, a/R*, e, i, omega   (planet_orbit)
  * transit: star with Claret 4-coefficient limb darkening
        I(mu) = 1 - sum_n a_n (1 - mu^(n/2)),  n = 1..4
    occulted by an opaque disk of radius p = Rp/R* at projected separation z:
        dF = int I(r) r theta(r) dr,   theta = arc of the radius-r circle inside the planet
    evaluated with a tanh-sinh rule (endpoint singularities of theta and of
    mu^(1/2) at the limb are integrable);
  * eclipse: fraction of the planet's disk hidden by the star (uniform disk).

On the exposure path the K x W depth matrix is produced by the k_lightcurve
HIP kernel from z[K], p[W] and the four coefficients (the same rule and
nodes, float32 integrand / float64 sum); the numpy version here serves
`Observation.show_lightcurve` and the tests.
"""
import numpy as np

N_NODES = 24      # 7e-11 absolute on the flux (tests/test_lightcurve.py); the device kernel uses the same rule


def tanh_sinh_nodes(n=N_NODES, t_max=3.0):
    """Nodes x_i in (0, 1) and weights w_i of the double-exponential rule
    int_0^1 f(x) dx ~ sum w_i f(x_i) (Takahasi & Mori 1974)."""
    t = np.linspace(-t_max, t_max, n)
    h = t[1] - t[0]
    u = 0.5 * np.pi * np.sinh(t)
    x = 0.5 * (1.0 + np.tanh(u))
    w = h * 0.25 * np.pi * np.cosh(t) / np.cosh(u) ** 2
    # distance of each node from its nearer end, kept accurately (x itself rounds to 0/1)
    d = 0.5 * np.exp(-np.abs(u)) / np.cosh(u)
    return x, w, d


_X, _W, _D = tanh_sinh_nodes()


def claret_intensity(mu, ld):
    s = np.sqrt(mu)
    a1, a2, a3, a4 = ld
    return 1.0 - a1 * (1 - s) - a2 * (1 - mu) - a3 * (1 - mu * s) - a4 * (1 - mu * mu)


def stellar_flux_total(ld):
    """int over the disk of I: pi (1 - sum a_n n/(n+4))."""
    a = np.asarray(ld, dtype=float)
    n = np.arange(1, 5)
    return np.pi * (1.0 - np.sum(a * n / (n + 4.0)))


def transit_flux(z, p, ld):
    """Normalised flux of the star during transit; z, p broadcastable arrays."""
    z, p = np.broadcast_arrays(np.asarray(z, dtype=float), np.asarray(p, dtype=float))
    out = np.ones(z.shape)
    f0 = stellar_flux_total(ld)
    touching = z < 1.0 + p
    if not np.any(touching):
        return out
    zz, pp = z[touching], p[touching]
    # part 1: radii fully inside the planet (only when the planet covers the centre): r < p - z
    r_full = np.clip(pp - zz, 0.0, 1.0)
    # int_0^{r_full} I(r) 2 pi r dr, analytic in mu: with mu_f = sqrt(1 - r_full^2)
    mu_f = np.sqrt(1.0 - r_full ** 2)

    def cum(mu):   # int_mu^1 I(m) 2 pi m dm
        a1, a2, a3, a4 = ld
        def prim(m):
            return (m * m / 2.) * (1 - a1 - a2 - a3 - a4) + a1 * m ** 2.5 / 2.5 + a2 * m ** 3 / 3. + \
                a3 * m ** 3.5 / 3.5 + a4 * m ** 4 / 4.
        return 2 * np.pi * (prim(1.0) - prim(mu))
    d_f = cum(mu_f)
    # part 2: partially covered radii |z - p| < r < min(1, z + p)
    ra = np.abs(zz - pp)
    rb = np.minimum(1.0, zz + pp)
    ok = rb > ra
    L = np.where(ok, rb - ra, 0.0)[:, None]
    # node positions measured from the nearer end so that differences stay accurate
    x = _X[None, :]
    r = ra[:, None] + L * x
    lo = L * np.where(x < 0.5, _D[None, :], 1 - _D[None, :])         # r - ra
    hi = L * np.where(x < 0.5, 1 - _D[None, :], _D[None, :])         # rb - r
    # theta = 2 acos((r^2 + z^2 - p^2) / (2 r z)) = 4 atan2(sqrt(p^2 - (r - z)^2), sqrt((r + z)^2 - p^2))
    zc, pc = zz[:, None], pp[:, None]
    num = np.maximum(pc ** 2 - (r - zc) ** 2, 0.0)
    den = np.maximum((r + zc) ** 2 - pc ** 2, 0.0)
    theta = 4.0 * np.arctan2(np.sqrt(num), np.sqrt(den))
    # 1 - r^2 accurately near the limb: (1 - r)(1 + r) with 1 - r = (1 - rb) + hi
    one_minus_r = (1.0 - rb)[:, None] + hi
    mu = np.sqrt(np.maximum(one_minus_r * (1.0 + r), 0.0))
    integrand = claret_intensity(mu, ld) * r * theta
    d_p = (integrand * _W[None, :]).sum(axis=1) * L[:, 0]
    out[touching] = 1.0 - (d_f + d_p) / f0
    return out


def uniform_overlap_fraction(z, p):
    """Area of the intersection of a unit disk and a disk of radius p at distance z, over pi p^2
    (the fraction of the PLANET hidden by / in front of the star)."""
    z, p = np.broadcast_arrays(np.asarray(z, dtype=float), np.asarray(p, dtype=float))
    out = np.zeros(z.shape)
    inside = z <= 1.0 - p
    out[inside] = 1.0
    part = (z > np.abs(1.0 - p)) & (z < 1.0 + p)
    zz, pp = z[part], p[part]
    k0 = np.arccos(np.clip((pp ** 2 + zz ** 2 - 1) / (2 * pp * zz), -1, 1))
    k1 = np.arccos(np.clip((1 - pp ** 2 + zz ** 2) / (2 * zz), -1, 1))
    area = pp ** 2 * k0 + k1 - 0.5 * np.sqrt(np.maximum(4 * zz ** 2 - (1 + zz ** 2 - pp ** 2) ** 2, 0.0))
    out[part] = area / (np.pi * pp ** 2)
    return out


def planet_position(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, time_array,
                    time_origin=0.0):
    """The planet's position in stellar radii at each time: (X, Y) on the sky, Z along the line of sight (> 0: in
    front of the star).  Times are time_origin + time_array; the sum is never formed (at JD-scale origins it would
    round small offsets to 4.7e-10 d: planet_orbit_short_span hands in an exposure's start and offsets from it)."""
    t = np.asarray(time_array, dtype=float)
    e = float(eccentricity)
    inc = np.radians(inclination_deg)
    w = np.radians(periastron_deg if np.isfinite(periastron_deg) else 0.0)
    f_tr = 0.5 * np.pi - w                                    # true anomaly at mid-transit
    E_tr = 2.0 * np.arctan(np.sqrt((1 - e) / (1 + e)) * np.tan(0.5 * f_tr))
    t_peri = mid_time - period * (E_tr - e * np.sin(E_tr)) / (2 * np.pi)
    M = 2 * np.pi * (((t + (time_origin - t_peri)) / period) % 1.0)
    E = M.copy()
    for _ in range(60):                                       # Kepler: Newton
        dE = (E - e * np.sin(E) - M) / (1 - e * np.cos(E))
        E = E - dE
        if np.max(np.abs(dE)) < 1e-14:
            break
    f = 2.0 * np.arctan2(np.sqrt(1 + e) * np.sin(E / 2), np.sqrt(1 - e) * np.cos(E / 2))
    r = sma_over_rs * (1 - e * e) / (1 + e * np.cos(f))
    X = -r * np.cos(w + f)
    Y = -r * np.sin(w + f) * np.cos(inc)
    Zlos = r * np.sin(w + f) * np.sin(inc)
    return X, Y, Zlos


def planet_orbit(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, time_array):
    """Projected star-planet separation z (stellar radii) and the planet's
    line-of-sight coordinate (> 0: in front of the star) at each time."""
    X, Y, Zlos = planet_position(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, time_array)
    return np.sqrt(X * X + Y * Y), Zlos


_CHEB_NODES = 10
_cheb_cache = {}


def _cheb_matrix(x):
    """Barycentric interpolation matrix from the _CHEB_NODES Chebyshev points of [-1, 1] to the abscissae x."""
    m = _CHEB_NODES
    j = np.arange(m)
    nodes = np.cos(np.pi * j / (m - 1))
    wts = (-1.0) ** j
    wts[0] *= 0.5
    wts[-1] *= 0.5
    d = x[:, None] - nodes[None, :]
    exact = d == 0.0
    d[exact] = 1.0
    B = wts[None, :] / d
    B /= B.sum(axis=1)[:, None]
    rows = exact.any(axis=1)
    B[rows] = exact[rows].astype(float)
    return nodes, B


def planet_orbit_short_span(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, time_array):
    """planet_orbit for the K sub-sample times of ONE exposure (thousands of them on a finely sampled scan: 2233 in
    the reference's example visit, the host's largest per-exposure cost at 0.15 ms).  Over an exposure the planet moves
    along ~1e-4 of its orbit and its position is an analytic function of time, so X, Y and Z are evaluated at ten
    Chebyshev points of the span and interpolated (barycentric form; the n-th term of their Taylor series falls like
    (2 pi span / period)^n / n!: the interpolant is exact to rounding, tests/test_lightcurve.py) -- the position, not
    the separation sqrt(X^2 + Y^2), which has a corner at mid-transit for an edge-on orbit.  Longer spans (more than
    0.4 % of the period: half an hour of a five-day orbit) or few samples take planet_orbit."""
    t = np.asarray(time_array, dtype=float)
    if t.size < 4 * _CHEB_NODES:
        return planet_orbit(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, t)
    t_lo, t_hi = float(t.min()), float(t.max())
    half = 0.5 * (t_hi - t_lo)
    if not (0.0 < half <= 0.002 * period):
        return planet_orbit(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, t)
    # the sub-sample times of every exposure of a visit are the same offsets from its start: one matrix serves them all.
    # Abscissae AND node times are built from the offsets and ONE centre, t[0] + off_mid: at JD-scale times (ulp 4.7e-10
    # d) a centre taken as (t_hi + t_lo) / 2 rounds differently from exposure to exposure, and a cached matrix built
    # around the first exposure's centre would then sit up to 1e-6 of the span off the nodes of the others.
    off = t - t[0]
    off_lo, off_hi = float(off.min()), float(off.max())
    half = 0.5 * (off_hi - off_lo)
    off_mid = 0.5 * (off_hi + off_lo)
    if not half > 0.0:
        return planet_orbit(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time, t)
    key = (t.size, off[1], off[-1], half)
    hit = _cheb_cache.get(key)
    if hit is None or not np.array_equal(hit[0], off):
        x = (off - off_mid) / half
        nodes, B = _cheb_matrix(x)
        if len(_cheb_cache) > 64:
            _cheb_cache.clear()
        hit = _cheb_cache[key] = (off.copy(), nodes, B)
    _, nodes, B = hit
    X, Y, Zlos = planet_position(period, sma_over_rs, eccentricity, inclination_deg, periastron_deg, mid_time,
                                 off_mid + half * nodes, time_origin=float(t[0]))
    P = B @ np.stack([X, Y, Zlos], axis=1)
    return np.sqrt(P[:, 0] * P[:, 0] + P[:, 1] * P[:, 1]), P[:, 2]


def transit(ld, rp_over_rs, period, sma_over_rs, eccentricity, inclination, periastron, mid_time, time_array):
    """Normalised stellar flux (the role of pylightcurve.transit('claret', ...))."""
    z, los = planet_orbit(period, sma_over_rs, eccentricity, inclination, periastron, mid_time, time_array)
    z = np.where(los > 0, z, 10.0 + z)                         # behind the star: no transit
    return transit_flux(z, rp_over_rs, ld)


def eclipse(fp_over_fs, rp_over_rs, period, sma_over_rs, eccentricity, inclination, periastron, mid_time,
            time_array):
    """Normalised star + planet flux through secondary eclipse (the role of pylightcurve.eclipse)."""
    z, los = planet_orbit(period, sma_over_rs, eccentricity, inclination, periastron, mid_time, time_array)
    hidden = np.where(los < 0, uniform_overlap_fraction(z, rp_over_rs), 0.0)
    return (1.0 + fp_over_fs * (1.0 - hidden)) / (1.0 + fp_over_fs)


def depth_inputs(period, sma_over_rs, eccentricity, inclination, periastron, mid_time, time_array, rp_white):
    """The K-vectors the device kernel needs: z for the transit (>= 10 when the planet is
    behind the star) and the hidden fraction of the planet for the eclipse term."""
    z, los = planet_orbit_short_span(period, sma_over_rs, eccentricity, inclination, periastron, mid_time, time_array)
    z_tr = np.where(los > 0, z, 10.0 + z)
    behind = los < 0
    hidden = np.zeros(z.shape)
    if behind.any():                     # (the eclipse side of the orbit only: in transit there is nothing to evaluate)
        hidden[behind] = uniform_overlap_fraction(z[behind], rp_white)
    return z_tr, hidden


def planet_depths(ld, planet_spectrum, z_tr, hidden):
    """1 - (transit - (1 - eclipse)) per sub-sample and wavelength bin: what
    Observation._generate_exposure hands to scanning_frame (observation.py:349-355, 442-443)."""
    p = np.sqrt(np.asarray(planet_spectrum, dtype=float))
    tr = transit_flux(np.asarray(z_tr)[:, None], p[None, :], ld)
    f = np.asarray(planet_spectrum, dtype=float)[None, :]
    ecl = (1.0 + f * (1.0 - np.asarray(hidden)[:, None])) / (1.0 + f)
    return 1.0 - (tr - (1.0 - ecl))


class DeviceDepths(object):
    """Hand this to ExposureGenerator.scanning_frame as `planet_signal` to have the
    K x W transit-depth matrix computed on the GPU (k_lightcurve) instead of passing it."""

    def __init__(self, z_tr, hidden, planet_spectrum, ld):
        self.z_tr = np.asarray(z_tr, dtype=float)
        self.hidden = None if hidden is None else np.asarray(hidden, dtype=float)
        self.planet_spectrum = np.asarray(planet_spectrum, dtype=float)
        self.ld = [float(v) for v in ld]

    def host_matrix(self):
        """The same matrix evaluated with numpy (for tests and plots)."""
        hidden = np.zeros_like(self.z_tr) if self.hidden is None else self.hidden
        return planet_depths(self.ld, self.planet_spectrum, self.z_tr, hidden)
