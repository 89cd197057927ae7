"""Visit-level (science-level) measurement: inject a transit-depth spectrum into a synthetic visit, generate the visit
through the HIP path in several modes, extract spectral light curves the way an observer would, and fit the depths.

What the reference exists for is recovering transit depths from simulated visits (observation.py:293-357 builds the
per-wavelength light curves, exposure_generator.py:344-348, 625-628 turns them into electrons); the question a user has
about the production thrower (multinomial narrow component, one-word lane electrons, float32 reads) is whether the
depths it delivers differ from those of the per-electron / float64 path -- in ppm of depth, not in pixels.

TEST INFRASTRUCTURE (shared by tests/test_visit_science_gpu.py and scripts/visit_science.py); the numpy extraction
below is the observer's side, nothing of the product.

The visit.  synthetic.Visit: depth d(lambda) = 0.0146 + 2e-4 sin(2 pi (lambda - 1.1 um) / 0.3 um)
times a smooth trapezoid g(t) over +-0.1 d; every detector effect of the configuration on (flat, sky, gain, dark,
non-linearity, clip, read noise, stellar Poisson noise, the visit's hook) except cosmic rays, which an observer rejects
before extracting and this extraction does not (ScienceVisit(cosmic_rays=True) keeps them: they cancel in a pair).  Star positions x_ref + phi_x,
y_ref + phi_y with independent uniform sub-pixel phases phi in [0, 1) per exposure: the sub-pixel phase is where a
position-rounding defect of a thrower would show.

The extraction (per exposure; calibration planes known, as an observer's pipeline knows its reference files):
  1. D_r = read_r - read_0 (removes the initial bias), linearised with the coefficient planes: L_r = D_r (1 + c1 + c2 D_r
     + c3 D_r^2 + c4 D_r^3) -- the exact inverse of the simulated non-linearity (detector.py:318-350);
  2. minus the super-dark of read r; differences of consecutive reads I_r = (L_r - dark_r) - (L_{r-1} - dark_{r-1}):
     the electrons of read interval r in DN;
  3. times the gain 2.35 / pfl -> electrons; the sky level of the exposure fitted on columns left of the spectrum with
     the master-sky template and subtracted;
  4. rows of interval r: the scan's rows in that interval +- 14 px; columns: channels of equal width in the STAR's
     frame (offsets from the exposure's x_ref, fractional weights on the two edge columns);
  5. flux of channel c = sum over r of the box sums ("up-the-ramp differences"), and, separately, the same box over the
     whole scan on the last read alone ("last read").
The fit: F_ic / hook_i = A_c (1 - delta_c G_i) by least squares, G_i the duration-weighted mean of g over the exposure's
sub-samples; delta_c's error from the residual scatter.  Injected: delta_c = sum_w W_cw d_w / sum_w W_cw with W_cw the
electrons bin w sends into channel c (stellar flux x sensitivity x bin width x the PSF's mass on the channel's columns).
Paired comparison of two modes (same stellar counts, same sky / dark / read-noise / cosmic-ray draws: the counters do not
depend on the thrower or on the reads' type): r_ic = F^a_ic / F^b_ic - 1 fitted the same way -> delta^a_c - delta^b_c
with the common noise gone.
"""
import numpy as np

from wayne_amd import _lib, engine, synthetic, tools
from wayne_amd.exposure_generator import ExposureGenerator

MODES = {
    # name: (rng_mode, reads' dtype, exact samplers)
    "production": (_lib.RNG_SPLIT, np.float32, False),        # what bench.py's `value`, the CLI and the API default run
    "split_f64": (_lib.RNG_SPLIT, np.float64, False),         # the same thrower, float64 reads: isolates the reads' type
    "per_electron": (_lib.RNG_PHILOX, np.float64, False),     # every electron thrown, float64 reads: the reference's shape
    "replay": (_lib.RNG_REPLAY, np.float64, True),            # the reference's own rand_r streams, bit-exact scatter
}
N_CHANNELS = 20
WL_RANGES = {"G141": (1.12, 1.65), "G102": (0.82, 1.13)}       # micron: where the grism's sensitivity is high
ROW_MARGIN = 14
BG_COLS = (6, 26)             # bordered columns used for the sky level (left of the first-order spectrum)


class ScienceVisit(object):
    """A synthetic visit with sub-pixel phases, its expected channel depths, and the extraction geometry."""

    def __init__(self, name, n_exposures, seed=1963, cosmic_rays=False):
        import helpers
        self.v = v = helpers.make_visit(name, n_exposures=n_exposures, seed=seed)
        # cosmic rays: an observer rejects them before extracting (a hit is ~5 sigma of a channel's photon noise); the
        # extraction here has no rejection step, so the visit is generated without them unless asked (their law is
        # tested in tests/test_extremes_gpu.py; in a PAIRED comparison they cancel exactly -- same counters, same hits)
        self.frame_overrides = {} if cosmic_rays else {"cosmic_rate": None}
        rng = np.random.RandomState(seed + 77)
        self.phase_x = rng.uniform(0.0, 1.0, n_exposures)
        self.phase_y = rng.uniform(0.0, 1.0, n_exposures)
        v.x_refs = v.cfg["x_ref"] + self.phase_x
        v.y_refs = v.cfg["y_ref"] + self.phase_y
        self.N = v.detector.light_sensitive_size(v.SUBARRAY) if hasattr(v.detector, "light_sensitive_size") else (
            1014 if v.SUBARRAY == 1024 else v.SUBARRAY)
        self.S = self.N + 10
        self.R = v.NSAMP - 1
        self.sub_scale = 0 if v.SUBARRAY == 1024 else 507 - v.SUBARRAY // 2
        gr = v.grism
        i0, i1 = tools.crop_spectrum_ind(gr.wl_limits[0], gr.wl_limits[1], v.wl)
        self.wl = v.wl[i0:i1]
        self.crop = (i0, i1)
        # the trace: offsets of every bin from the star, in x (the field dependence over one pixel is < 1e-4 px)
        tr = gr.get_trace(v.cfg["x_ref"] + 0.5, v.cfg["y_ref"] + 0.5)
        self.dx = np.asarray(tr.wl_to_x(self.wl), dtype=float) - (v.cfg["x_ref"] + 0.5)
        self.dy = np.asarray(tr.wl_to_y(self.wl), dtype=float) - (v.cfg["y_ref"] + 0.5)
        lo_wl, hi_wl = WL_RANGES[gr.name]
        edges_wl = np.linspace(lo_wl, hi_wl, N_CHANNELS + 1)
        self.edges = np.interp(edges_wl, self.wl, self.dx)               # channel edges as offsets from the star, px
        self.channel_wl = 0.5 * (edges_wl[1:] + edges_wl[:-1])
        # transit
        self.read_times = np.asarray(v.read_times, dtype=float)
        t_sub = v.exp_start_days[:, None] + v.sample_mid_points[None, :] / 86400e3
        g = synthetic.transit_shape(t_sub)
        self.G = (g * v.sample_durations[None, :]).sum(axis=1) / v.sample_durations.sum()
        self.hook = np.array([v.scale_factor(i) for i in range(n_exposures)])
        # calibration planes of the mode, bordered
        planes = v.calibration.for_mode(gr.name, v.SUBARRAY, v.SAMPSEQ, self.read_times, detector=v.detector)
        self.lin = [np.asarray(p, dtype=np.float64) for p in planes["lin"]]
        self.dark = np.concatenate([np.zeros((1, self.S, self.S)), np.asarray(planes["dark_sci"], dtype=np.float64)])
        gain = np.full((self.S, self.S), 2.35)
        gain[5:-5, 5:-5] = 2.35 / np.asarray(planes["pfl"], dtype=np.float64)
        self.gain = gain
        sky = np.zeros((self.S, self.S))
        sky[5:-5, 5:-5] = np.asarray(planes["sky"], dtype=np.float64)
        self.sky_template = sky
        self.dt = np.diff(np.concatenate([[0.0], self.read_times]))
        self.expected = self._expected_depths()

    # -- what was injected, per channel -------------------------------------------------------------------------
    def _expected_depths(self):
        from scipy.special import ndtr
        v, gr = self.v, self.v.grism
        i0, i1 = self.crop
        sens_wl, sens_val = v.calibration.sensitivity(gr.name)
        weight = v.stellar_flux[i0:i1] * np.interp(self.wl, sens_wl, sens_val) * tools.bin_centers_to_widths(self.wl)
        ratio = np.clip(np.polyval(gr.psf_ratio_poly, self.wl), 0.0, 1.0)        # the WIDE fraction (pyparallel_menu.c:89)
        sl, sh = np.polyval(gr.psf_sigmal_poly, self.wl), np.polyval(gr.psf_sigmah_poly, self.wl)
        d = v.depth0[i0:i1]
        out = np.empty(N_CHANNELS)
        self.channel_electrons = np.empty(N_CHANNELS)
        # the wavelength solution is field dependent (grism.py:779-803): along a 700-row scan a wavelength's column moves
        # by a fraction of a pixel against the star, so the PSF's mass on a channel is averaged over the scan
        # (9 positions; the channel EDGES stay those of the scan's start: they are the observer's choice)
        tr0 = (v.cfg["x_ref"] + 0.5, v.cfg["y_ref"] + 0.5)
        dxs = [np.asarray(gr.get_trace(tr0[0], tr0[1] + v.scan_speed * t).wl_to_x(self.wl), dtype=float) - tr0[0]
               for t in np.linspace(0.0, self.read_times[-1], 9)] if v.scan_speed else [self.dx]
        for c in range(N_CHANNELS):
            a, b = self.edges[c], self.edges[c + 1]
            mass = np.mean([ratio * (ndtr((b - dx) / sh) - ndtr((a - dx) / sh)) +
                            (1 - ratio) * (ndtr((b - dx) / sl) - ndtr((a - dx) / sl)) for dx in dxs], axis=0)
            w = weight * mass
            out[c] = (w * d).sum() / w.sum()
            self.channel_electrons[c] = w.sum() * self.read_times[-1] * 1e4 * 1e-3 * 1e3   # (flux x sens x dlam[um] 1e4 x s)
        self.white_expected = float((self.channel_electrons * out).sum() / self.channel_electrons.sum())
        return out

    # -- the observer's extraction ------------------------------------------------------------------------------
    def column_weights(self, i):
        """(N_CHANNELS, S) weights of the bordered columns for exposure i: channel c covers [x* + e_c, x* + e_{c+1})."""
        x_star = self.v.x_refs[i] - self.sub_scale + 5.0          # bordered column coordinate of the star
        cols = np.arange(self.S, dtype=float)
        w = np.empty((N_CHANNELS, self.S))
        for c in range(N_CHANNELS):
            lo, hi = x_star + self.edges[c], x_star + self.edges[c + 1]
            w[c] = np.clip(np.minimum(cols + 1.0, hi) - np.maximum(cols, lo), 0.0, 1.0)
        return w

    def row_window(self, i, t0_s, t1_s):
        """Bordered rows the spectrum crosses between t0 and t1 of exposure i, +- ROW_MARGIN."""
        v = self.v
        y0 = v.y_refs[i] - self.sub_scale + 5.0 + self.dy.min()
        y1 = v.y_refs[i] - self.sub_scale + 5.0 + self.dy.max()
        lo = int(np.floor(y0 + v.scan_speed * t0_s)) - ROW_MARGIN
        hi = int(np.ceil(y1 + v.scan_speed * t1_s)) + ROW_MARGIN + 1
        return max(lo, 5), min(hi, self.S - 5)

    def extract(self, i, reads):
        """reads (R + 1, S, S) -> (flux by up-the-ramp differences [N_CHANNELS], flux from the last read [N_CHANNELS])."""
        reads = np.asarray(reads, dtype=np.float64)
        D = reads[1:] - reads[0]
        c1, c2, c3, c4 = self.lin
        L = D * (1.0 + c1 + D * (c2 + D * (c3 + c4 * D)))
        L -= self.dark[1:]
        I = np.diff(np.concatenate([np.zeros((1, self.S, self.S)), L]), axis=0) * self.gain      # electrons per interval
        cw = self.column_weights(i)
        ramp = np.zeros(N_CHANNELS)
        t_prev = 0.0
        for r in range(self.R):
            r0, r1 = self.row_window(i, t_prev, self.read_times[r])
            img = I[r, r0:r1]
            T = self.sky_template[r0:r1] * self.dt[r]
            b0, b1 = BG_COLS
            s = img[:, b0:b1].sum() / T[:, b0:b1].sum()                 # sky level, electrons per second per unit template
            ramp += cw @ (img - s * T).sum(axis=0)
            t_prev = self.read_times[r]
        # the last read alone: one box over the whole scan
        r0, r1 = self.row_window(i, 0.0, self.read_times[-1])
        img = L[-1, r0:r1] * self.gain[r0:r1]
        T = self.sky_template[r0:r1] * self.read_times[-1]
        b0, b1 = BG_COLS
        s = img[:, b0:b1].sum() / T[:, b0:b1].sum()
        last = cw @ (img - s * T).sum(axis=0)
        return ramp, last


def generate(sv, mode, indices=None, depth=3, depth_scale=1.0):
    """Flux tables (n, N_CHANNELS) x 2 of the visit in `mode`, exposures pipelined over `depth` context slots.
    `depth_scale`: the injected transit depths multiplied by it (the estimator's own injection-recovery check)."""
    v = sv.v
    rng_mode, out_dtype, exact = MODES[mode]
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    idx = list(range(v.n_exposures)) if indices is None else list(indices)
    ramp = np.empty((len(idx), N_CHANNELS))
    last = np.empty((len(idx), N_CHANNELS))
    in_flight = []

    def finish():
        n, i, slot = in_flight.pop(0)
        reads = ctx.wait(slot)
        ramp[n], last[n] = sv.extract(i, reads)

    for n, i in enumerate(idx):
        eg = ExposureGenerator(v.detector, v.grism, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=v.calibration,
                               seed=v.seed, exposure_index=i)
        over = dict(sv.frame_overrides)
        if depth_scale != 1.0:
            over["planet_signal"] = v.planet_signal(i) * depth_scale
        desc = eg.build_descriptor(eng, rng_mode=rng_mode, out_dtype=out_dtype, exact_samplers=exact, threads=2,
                                   **v.frame_kwargs(i, **over))
        if len(in_flight) >= depth:
            finish()
        slot = n % (depth + 1)
        ctx.upload(slot, desc)
        ctx.run(slot)
        ctx.fetch_async(slot)
        in_flight.append((n, i, slot))
    while in_flight:
        finish()
    return ramp, last


def fit_depths(flux, G, hook=None):
    """Least squares of flux_ic / hook_i = A_c (1 - delta_c G_i) -> (delta [C], sigma_delta [C], relative residual rms [C])."""
    y = np.asarray(flux, dtype=float)
    if hook is not None:
        y = y / np.asarray(hook)[:, None]
    X = np.column_stack([np.ones_like(G), G])
    beta, _, _, _ = np.linalg.lstsq(X, y, rcond=None)
    a, b = beta
    res = y - X @ beta
    dof = len(G) - 2
    s2 = (res ** 2).sum(axis=0) / dof
    cov = np.linalg.inv(X.T @ X)
    delta = -b / a
    # var(-b/a) to first order
    var = s2 * (cov[1, 1] / a ** 2 + b ** 2 * cov[0, 0] / a ** 4 - 2 * b * cov[0, 1] / a ** 3)
    return delta, np.sqrt(var), np.sqrt(s2) / a


def fit_paired(flux_a, flux_b, G):
    """delta_a - delta_b per channel from r_ic = F^a_ic / F^b_ic - 1 = alpha_c - (delta^a_c - delta^b_c) G_i / (1 - delta G_i):
    the common noise (stellar counts, sky, read noise, cosmic rays: same counters in both modes) divides out."""
    r = np.asarray(flux_a, dtype=float) / np.asarray(flux_b, dtype=float) - 1.0
    X = np.column_stack([np.ones_like(G), G])
    beta, _, _, _ = np.linalg.lstsq(X, r, rcond=None)
    res = r - X @ beta
    s2 = (res ** 2).sum(axis=0) / (len(G) - 2)
    cov = np.linalg.inv(X.T @ X)
    return -beta[1], np.sqrt(s2 * cov[1, 1]), np.sqrt(s2)


def phase_trend(values, phase):
    """Dependence of `values` (n,) on a sub-pixel phase in [0, 1): least squares of a + b cos 2 pi phi + c sin 2 pi phi
    -> ((b, c), their errors, chi2 of b = c = 0 with 2 degrees of freedom).  A thrower that rounded positions on the
    frame's scale, or dropped the fraction of a pixel, would show here as a first harmonic."""
    phi = 2.0 * np.pi * np.asarray(phase, dtype=float)
    X = np.column_stack([np.ones_like(phi), np.cos(phi), np.sin(phi)])
    beta, _, _, _ = np.linalg.lstsq(X, values, rcond=None)
    res = values - X @ beta
    s2 = (res ** 2).sum() / (len(values) - 3)
    cov = s2 * np.linalg.inv(X.T @ X)
    bc, cbc = beta[1:], cov[1:, 1:]
    return bc, np.sqrt(np.diag(cbc)), float(bc @ np.linalg.solve(cbc, bc))


def white(flux, weights=None):
    return np.asarray(flux).sum(axis=1, keepdims=True)


def analyse(sv, tables, subset=None):
    """tables: {mode: (ramp, last)}; subset: {mode: indices} for modes generated on part of the visit.
    -> the JSON-able report: absolute recovery per mode, paired differences against `per_electron`, phase trends."""
    subset = subset or {}
    out = {"config": sv.v.name, "n_exposures": int(sv.v.n_exposures), "channels_um": [round(float(x), 4) for x in sv.channel_wl],
           "injected_ppm": [round(float(x) * 1e6, 2) for x in sv.expected],
           "photon_noise_ppm_per_exposure": [round(float(1e6 / np.sqrt(e)), 1) for e in sv.channel_electrons],
           "modes": {}, "paired": {}}
    for mode, (ramp, last) in tables.items():
        idx = np.asarray(subset.get(mode, np.arange(sv.v.n_exposures)))
        G, hook = sv.G[idx], sv.hook[idx]
        rep = {}
        for how, flux in (("ramp", ramp), ("last_read", last)):
            d, s, rms = fit_depths(flux, G, hook)
            dw, sw, rmsw = fit_depths(white(flux), G, hook)
            pull = (d - sv.expected) / s
            rep[how] = {"recovered_minus_injected_ppm": [round(float(x) * 1e6, 2) for x in d - sv.expected],
                        "sigma_ppm": [round(float(x) * 1e6, 2) for x in s],
                        "pull": [round(float(x), 2) for x in pull], "chi2": float((pull ** 2).sum()), "dof": N_CHANNELS,
                        "residual_rms_over_photon_noise": [round(float(r * np.sqrt(e)), 3) for r, e in zip(rms, sv.channel_electrons)],
                        "white_recovered_minus_injected_ppm": round(float(dw[0] - sv.white_expected) * 1e6, 3),
                        "white_sigma_ppm": round(float(sw[0]) * 1e6, 3)}
        rep["n"] = int(len(idx))
        out["modes"][mode] = rep
    ref = "per_electron"
    for mode in tables:
        if mode == ref or ref not in tables:
            continue
        idx = np.asarray(subset.get(mode, np.arange(sv.v.n_exposures)))
        pos = {int(i): n for n, i in enumerate(np.asarray(subset.get(ref, np.arange(sv.v.n_exposures))))}
        sel = np.array([pos[int(i)] for i in idx])
        rep = {}
        for how, k in (("ramp", 0), ("last_read", 1)):
            fa, fb = tables[mode][k], tables[ref][k][sel]
            d, s, rms = fit_paired(fa, fb, sv.G[idx])
            dw, sw, rmsw = fit_paired(white(fa), white(fb), sv.G[idx])
            r_white = (white(fa) / white(fb) - 1.0)[:, 0]
            r_chan = (fa / fb - 1.0).mean(axis=1)
            px = phase_trend(r_chan, sv.phase_x[idx])
            py = phase_trend(r_chan, sv.phase_y[idx])
            # ... and channel by channel in x, where a shift of the spectrum against the star-fixed channels shows first (the
            # steep flanks of the sensitivity curve): the largest chi2 (2 dof each) of the 20
            by_channel = [phase_trend((fa / fb - 1.0)[:, c], sv.phase_x[idx])[2] for c in range(N_CHANNELS)]
            # the static part of the pair: does one mode put a channel's electrons into its neighbours?  Mean flux ratio
            # per channel over the visit (a transit-independent redistribution cancels in a depth; it shows here)
            rc = fa / fb - 1.0
            off, off_s = rc.mean(axis=0), rc.std(axis=0, ddof=1) / np.sqrt(rc.shape[0])
            rep[how] = {"depth_difference_ppm": [round(float(x) * 1e6, 3) for x in d],
                        "sigma_ppm": [round(float(x) * 1e6, 3) for x in s],
                        "chi2": float(((d / s) ** 2).sum()), "dof": N_CHANNELS,
                        "paired_flux_rms_ppm": [round(float(x) * 1e6, 2) for x in rms],
                        "white_depth_difference_ppm": round(float(dw[0]) * 1e6, 4), "white_sigma_ppm": round(float(sw[0]) * 1e6, 4),
                        "white_flux_offset_ppm": round(float(r_white.mean()) * 1e6, 4),
                        "white_flux_offset_sigma_ppm": round(float(r_white.std(ddof=1) / np.sqrt(len(r_white))) * 1e6, 4),
                        "flux_ratio_vs_x_phase_by_channel_chi2": [round(float(x), 2) for x in by_channel],
                        "channel_flux_offset_ppm": [round(float(x) * 1e6, 2) for x in off],
                        "channel_flux_offset_sigma_ppm": [round(float(x) * 1e6, 2) for x in off_s],
                        "flux_ratio_vs_x_phase_ppm": {"cos_sin": [round(float(x) * 1e6, 3) for x in px[0]],
                                                      "err": [round(float(x) * 1e6, 3) for x in px[1]], "chi2": px[2], "dof": 2},
                        "flux_ratio_vs_y_phase_ppm": {"cos_sin": [round(float(x) * 1e6, 3) for x in py[0]],
                                                      "err": [round(float(x) * 1e6, 3) for x in py[1]], "chi2": py[2], "dof": 2}}
        rep["n"] = int(len(idx))
        out["paired"]["%s_minus_%s" % (mode, ref)] = rep
    return out
