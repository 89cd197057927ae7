"""GPU: the PRODUCTION thrower modes (WAYNE_RNG_SPLIT in production math -- flags = 0 -- and WAYNE_RNG_PHILOX) against
the reference's compiled C thrower IN DISTRIBUTION.

The replay mode reproduces the reference's frames bit for bit (tests/test_psf_gpu.py); the modes `bench.py` times
cannot -- the reference's frame is one sample of a law, theirs another sample of (what must be) the same law.  So:
ensembles.  For each input, M frames of `oracle/_ref` (wayne/pyparallel_menu.c compiled unmodified; different `test`
seeds, `threads` 1 and 4 alternating) beside M' frames of each device mode through the C ABI (wayne_psf_apply), and both
beside the exact per-pixel moments of the reference's algorithm (tests/ensemble_stats.py: sums of binomials, closed
form).  Every approximation of the production path -- float32 Box-Muller on the hardware's log2 / sqrt / sin / cos,
one 32-bit word per electron (16-bit radius with a refined far cell, 23-bit angle), the Chebyshev fit of erfc, the
6.5 sigma cut and +-6 px window of the multinomial chains, float32 cell masses, pooled rows of 16 bins -- is inside
these frames; the tests see their SUM.

Inputs (all from the reference's own golden vectors, tests/golden/psf_*.npz):
  bright   s256_t4 counts x 60: 9.7e6 electrons, 2165 per bin -> nearly every bin's narrow component is a multinomial
           draw (k_narrow, pooled rows), its wide component lane-thrown (k_lane)
  thin     s256_t4 counts clipped below 32 and thinned: every electron is thrown one by one by its bin's lane
  edge     edge_low counts x 20: a trace along pixel row 1 that runs off the frame's left edge (x from -6): 38 % of
           the electrons are lost, row 0 / column 0 must stay empty

Bands are 5 standard errors of each figure's known sampling distribution (ensemble_stats.check / check_moments), plus
the small stated floors for the non-gaussianity of counts.  The measured values of every figure are written to
gpurun_out/ensemble_parity.json by the tests themselves (committed copy: profiles/r04/ensemble_parity.json).

Then the same at exposure level (small256: SUBARRAY 256, NSAMP 4, 9 sub-samples, 3e6 electrons): production
exposures (split thrower, float32 reads, hardware-math samplers, Philox streams) beside exposures of the oracle driven
by the reference's C thrower and numpy's legacy generator in the reference's call order
(`ExposureOracle(thrower="ref", draws=LegacyDraws)`): per-pixel mean and variance of the last read inside the
trace (stellar Poisson + scatter + flat + gain + non-linearity + dark + read noise), outside it (sky + dark + read
noise: exposure_generator.py:488-495, detector.py:185-198) and in the reference pixels (zero read + read noise).
"""
import json
import os

import numpy as np
import pytest
from scipy.special import polygamma

import ensemble_stats as es
import extreme_stats as xs
import helpers
from conftest import load_golden_psf
from oracle import clib
from oracle import wayne_oracle as wo
from wayne_amd import _lib

pytestmark = pytest.mark.gpu

CASES = {
    #          golden      scale  clip  M_ref  M_dev
    "bright": ("s256_t4",  60,    None, 48,    256),
    "thin":   ("s256_t4",  1,     31,   400,   1200),
    "edge":   ("edge_low", 20,    None, 64,    400),
}
_cache = {}
# WAYNE_ENSEMBLE_SCALE=n: n times as many frames on both sides (the bands are standard errors, so they tighten by
# sqrt(n)); the figures then go to ensemble_parity_x<n>.json.  profiles/r04/ensemble_parity_x8.json is such a run.
SCALE = max(1, int(os.environ.get("WAYNE_ENSEMBLE_SCALE", "1")))
REF_SEED = int(os.environ.get("WAYNE_ENSEMBLE_REF_SEED", "5"))      # which `test` seeds the reference frames take
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                      "ensemble_parity.json" if SCALE == 1 else "ensemble_parity_x%d.json" % SCALE)


def report(key, **figures):
    """Keep the measured figures next to the bands (gpurun_out/ensemble_parity.json -> profiles/rNN/)."""
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        d = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        d[key] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in figures.items()}
        json.dump(d, open(REPORT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def case_inputs(name):
    golden, scale, clip, m_ref, m_dev = CASES[name]
    k = load_golden_psf(golden)
    counts = k["counts"].astype(np.int64) * scale
    if clip is not None:
        counts = np.minimum(counts, clip)
        counts[::3] //= 4
    return k, counts.astype(np.int32), m_ref * SCALE, m_dev * SCALE


def reference_ensemble(name):
    """M frames of the reference's compiled C thrower, with the exact moments of its law (cached per session)."""
    if name not in _cache:
        if not clib.have_ref():
            pytest.skip("oracle/_ref not built")
        k, counts, m_ref, _ = case_inputs(name)
        n = k["nr"]
        tests = np.random.RandomState(REF_SEED).randint(0, 100000, m_ref)        # exposure_generator.py:327
        A = np.stack([clib.psf_reference(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"], n, n, int(tests[m]),
                                         1 if m % 2 == 0 else 4).reshape(n, n) for m in range(m_ref)])
        _cache[name] = (A, es.analytic_moments(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"], n))
    return _cache[name]


def device_ensemble(ctx, name, mode):
    k, counts, _, m_dev = case_inputs(name)
    n = k["nr"]
    return np.stack([ctx.psf_apply(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"], n, n, 1963 + m // 64,
                                   rng_mode=mode, exposure=m, subsample=m % 5).reshape(n, n) for m in range(m_dev)])


def extreme_figures(A, mean, var, s, label, inputs, frames=64):
    """The LARGEST deviations of an ensemble against the exact law (`z_max` was computed by
    ensemble_stats and never asserted): (a) the largest standardised pixel mean, against the normal extreme-value law of
    n_bright pixels -- with room for the skewness of a mean of M counts whose sum is >= 400 (Cornish-Fisher: +0.2 at 5
    sigma); (b) every pixel count of the first `frames` frames, one by one, against the law of that pixel: a sum of
    independent binomials, one per bin and component (pyparallel_menu.c:87-108) -- exact tails by the lattice saddlepoint
    formula for the pixels with at least 20 expected electrons (extreme_stats.poisson_binomial_tails): no count may be one the
    law does not produce (family-wise 1e-3) and the tails at 1e-4 / 1e-5 must be populated as the law says, on both sides;
    below 20 electrons the Poisson law of the pixel's mean, which bounds both tails of the true one.  The reference's
    own frames are put through the same function: what pins the law's restatement, tails included, to the reference."""
    from scipy import stats
    k_in, counts = inputs
    bad = []
    n = max(s["n_bright"], 1)
    z_bound = float(stats.norm.isf(1e-3 / (2.0 * n))) + 0.35
    if s["z_max"] > z_bound:
        bad.append("%s: largest standardised pixel mean %.2f (bound %.2f for %d pixels)" % (label, s["z_max"], z_bound, n))
    live = mean > 1e-12
    F = np.asarray(A[:frames], dtype=np.float64)
    if F[:, ~live].any():
        bad.append("%s: electrons where the law puts none" % label)
    rng = np.random.default_rng(12)
    dim = live & (mean < 20.0)
    t_dim = xs.poisson_tails(F[:, dim], np.broadcast_to(mean[dim], F[:, dim].shape), rng)
    by, bx = np.nonzero(mean >= 20.0)
    kb = F[:, by, bx]                                         # [frame][bright pixel]
    zb = (kb - mean[by, bx]) / np.sqrt(var[by, bx])
    cache = {}

    def terms_of(j):
        px = j % by.size
        if px not in cache:
            cache[px] = xs.thrower_pixel_terms(counts[None, :].astype(np.float64), k_in["x"][None, :], k_in["y"][None, :],
                                               k_in["ratio"], k_in["sl"], k_in["sh"], bx[px], by[px])
        return cache[px]

    t_bright = xs.poisson_binomial_tails(kb.ravel(), zb.ravel(), terms_of, rng)
    # (below 20 e- a thinly populated bin can still put a third of its electrons into one pixel: there the Poisson law
    # is the BOUND of both tails it always is, not the law -- impossible draws and too-heavy tails only)
    bad += xs.check(t_dim, label + " (per frame and pixel, below 20 e-)", exact_frequencies=False)
    bad += xs.check(t_bright, label + " (per frame and pixel, from 20 e-)")
    fig = dict(z_max=s["z_max"], z_max_bound=z_bound, largest_abs_z_per_frame=float(np.abs(zb).max()) if zb.size else 0.0)
    fig.update({"dim_" + k_: v_ for k_, v_ in xs.summary(t_dim).items()})
    fig.update({"bright_" + k_: v_ for k_, v_ in xs.summary(t_bright).items()})
    return bad, fig


@pytest.mark.parametrize("name", list(CASES))
def test_reference_ensemble_follows_the_exact_moments(name):
    # pins tests/ensemble_stats.analytic_moments (truncation, row / column 0, the deterministic sigma split) to the
    # reference itself before anything is compared with it
    A, (mean, var, var_other, _) = reference_ensemble(name)
    s = es.compare_with_moments(A, mean, var, var_other)
    k, _, _, _ = case_inputs(name)
    w = es.wings_against_moments(A, mean, var, k["x"], k["y"])
    report("psf/%s/reference_vs_exact_moments" % name, **dict(s, **w))
    bad = es.check_moments(s) + es.check_wings_against_moments(w)
    # the extreme-value figures of the REFERENCE's own frames: the yardstick the device's are held to below
    bad_x, fig = extreme_figures(A, mean, var, s, "reference C", case_inputs(name)[:2])
    report("psf/%s/reference_extremes" % name, **fig)
    assert not bad + bad_x, "; ".join(bad + bad_x)


@pytest.mark.parametrize("mode", [_lib.RNG_SPLIT, _lib.RNG_PHILOX], ids=["split", "philox"])
@pytest.mark.parametrize("name", list(CASES))
def test_production_thrower_against_reference_ensemble(gpu_ctx, name, mode):
    k, counts, _, _ = case_inputs(name)
    A, (mean, var, var_other, _) = reference_ensemble(name)
    B = device_ensemble(gpu_ctx, name, mode)
    assert B.min() >= 0
    # two-sample: device ensemble beside the reference's
    two = es.compare(B, A, k["x"], k["y"])
    bad = es.check(two, require_subpoisson=0.95 if name != "thin" else None)
    # one-sample: device ensemble against the exact moments of the reference's law (the sharper test: M' >> M)
    one = es.compare_with_moments(B, mean, var, var_other)
    one.update(es.wings_against_moments(B, mean, var, k["x"], k["y"]))      # the wings once more, one-sample
    bad += es.check_moments(one) + es.check_wings_against_moments(one)
    tag = "split" if mode == _lib.RNG_SPLIT else "philox"
    report("psf/%s/%s_vs_reference" % (name, tag), **two)
    report("psf/%s/%s_vs_exact_moments" % (name, tag), **one)
    # the largest single-pixel deviations, held to the same bounds as the reference's own frames
    bad_x, fig = extreme_figures(B, mean, var, one, tag, (k, counts))
    report("psf/%s/%s_extremes" % (name, tag), **fig)
    bad += bad_x
    if name == "bright":
        # the variance law is the reference's deterministic split N = (int)(counts * ratio) (pyparallel_menu.c:89),
        # not a per-electron (or per-pixel Poisson) one: in the core of the trace the two differ by > 4 %
        assert one["n_split"] >= 200 and one["other_over_exact"] > 1.03
        off = (1.0 - one["split_ratio_other"]) / one["split_se"]
        if off < 5.0:
            bad.append("core variance does not exclude the random-split law: %.1f sigma" % off)
    assert not bad, "%s / mode %d: %s\n%r\n%r" % (name, mode, "; ".join(bad), two, one)


def test_split_and_philox_ensembles_agree_with_each_other(gpu_ctx):
    # product against product on a fourth input (the full-array golden, x 2): two large ensembles, tight bands
    k = load_golden_psf("s1014_t4")
    counts = (k["counts"].astype(np.int64) * 2).astype(np.int32)
    n = k["nr"]
    x0, x1 = int(k["x"].min()) - 40, int(k["x"].max()) + 40
    y0, y1 = int(k["y"].min()) - 40, int(k["y"].max()) + 40
    ens = {}
    for mode in (_lib.RNG_SPLIT, _lib.RNG_PHILOX):
        fr = []
        for m in range(96 * SCALE):
            f = gpu_ctx.psf_apply(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"], n, n, 77, rng_mode=mode,
                                  exposure=m, subsample=3).reshape(n, n)
            assert f.sum() == f[y0:y1, x0:x1].sum() == counts.sum()      # reach: nothing beyond 6.9 sigma_h
            fr.append(f[y0:y1, x0:x1].copy())
        ens[mode] = np.stack(fr)
    side = max(ens[_lib.RNG_SPLIT].shape[1:])
    pad = [np.pad(e, ((0, 0), (0, side - e.shape[1]), (0, side - e.shape[2]))) for e in ens.values()]
    s = es.compare(pad[0], pad[1], k["x"] - x0, k["y"] - y0)
    s["row0"] = s["col0"] = 0.0          # cropped window: its first row / column are ordinary pixels
    report("psf/s1014_t4_x2/split_vs_philox", **s)
    bad = es.check(s)
    assert not bad, "; ".join(bad) + "\n%r" % s


# ---------------------------------------------------------------------------------------------------------------
# exposure level
# ---------------------------------------------------------------------------------------------------------------
def _exposure_ensembles(m_dev=200 * SCALE, m_ref=100 * SCALE, visit="small256", threads=(1, 2)):
    key = ("exposures", visit, m_dev, m_ref)
    if key not in _cache:
        if not clib.have_ref():
            pytest.skip("oracle/_ref not built")
        v = helpers.make_visit(visit)
        off = dict(cosmic_rate=None)      # 15 hits of 10-35 ke per exposure would swamp every pixel variance; the
        #                                   cosmic-ray statistics have their own test (tests/test_configs_gpu.py)
        kw = v.frame_kwargs(0, **off)
        dev = []
        for m in range(m_dev):
            pg = helpers.product_generator(v, m)                 # exposure index m: independent streams
            dev.append(np.array(pg.scanning_frame(out_dtype=np.float32, **kw).reads[-1][0], dtype=np.float64))
        eo = helpers.oracle_generator(v)
        okw = helpers.oracle_kwargs(kw)
        ref = [eo.scanning_frame(threads=threads[m % 2], draws=wo.LegacyDraws(4000 + m), thrower="ref",
                                 **okw)[-1] for m in range(m_ref)]
        # where the star's electrons land, from a run of its own with every noise source off (a mask taken from either
        # ensemble would select on that ensemble's noise and bias the comparison): last read less the initial bias frame
        quiet = dict(add_stellar_noise=False, sky_background=0.0, add_dark=False, add_read_noise=False, cosmic_rate=None)
        star = eo.scanning_frame(threads=threads[1], draws=wo.LegacyDraws(1), thrower="ref",
                                 **helpers.oracle_kwargs(v.frame_kwargs(0, **quiet)))[-1] - eo._gen_zero_read(True)
        _cache[key] = (v, np.stack(dev), np.stack(ref), star)
    return _cache[key]


def _region_stats(D, R, sel):
    """Two-sample per-pixel figures of float frames over the pixels `sel`."""
    Md, Mr = D.shape[0], R.shape[0]
    md, mr = D.mean(axis=0)[sel], R.mean(axis=0)[sel]
    vd, vr = D.var(axis=0, ddof=1)[sel], R.var(axis=0, ddof=1)[sel]
    z = (md - mr) / np.sqrt(vd / Md + vr / Mr)
    n = int(sel.sum())
    nu = (1.0 / Md + 1.0 / Mr) ** 2 / (1.0 / (Md * Md * (Md - 1.0)) + 1.0 / (Mr * Mr * (Mr - 1.0)))
    lr = np.log(vd / vr) - es.log_s2_bias(Md) + es.log_s2_bias(Mr)
    return dict(n=n, z_mean=float(z.mean()), z_mean_se=1.0 / np.sqrt(n), z_std=float(z.std(ddof=1)),
                z_std_expect=float(np.sqrt(nu / (nu - 2.0))), z_std_se=float(1.0 / np.sqrt(2.0 * n)),
                log_var=float(lr.mean()), log_var_se=float(np.sqrt(polygamma(1, (Md - 1) / 2.0) + polygamma(1, (Mr - 1) / 2.0)) / np.sqrt(n)),
                mean_d=float(md.mean()), mean_r=float(mr.mean()), var_d=float(vd.mean()), var_r=float(vr.mean()))


def test_production_exposures_against_reference_driven_oracle_ensemble():
    v, D, R, star = _exposure_ensembles()
    S = D.shape[1]
    interior = np.zeros((S, S), dtype=bool)
    interior[5:-5, 5:-5] = True
    inside = interior & (star > 30.0)            # DN of starlight in the last read
    outside = interior & (star < 0.5)
    # the read-noise-only pixels: the 5-px reference border
    border = ~interior
    assert inside.sum() > 4000 and outside.sum() > 40000
    bad = []
    for name, sel, jitter_floor in (("inside", inside, 0.01), ("outside", outside, 0.0), ("border", border, 0.0)):
        s = _region_stats(D, R, sel)
        report("exposure/small256/%s" % name, **s)
        # the per-sub-sample pointing jitter (0.025 px, exposure_generator.py:328-329) moves neighbouring pixels of the
        # trace together, so their z are correlated: the floor on the mean allows for that inside the trace only
        if abs(s["z_mean"]) > 5.0 * s["z_mean_se"] + jitter_floor:
            bad.append("%s: pixel means differ, mean z %.4f (se %.4f)" % (name, s["z_mean"], s["z_mean_se"]))
        if abs(s["z_std"] - s["z_std_expect"]) > 5.0 * s["z_std_se"] + 0.02:
            bad.append("%s: spread of z %.4f, expected %.4f" % (name, s["z_std"], s["z_std_expect"]))
        if abs(s["log_var"]) > 5.0 * s["log_var_se"] + 0.01:
            bad.append("%s: pixel variances differ, mean log ratio %.4f (se %.4f)" % (name, s["log_var"],
                                                                                    s["log_var_se"]))
    # the background's variance is what the reference's stages say it is (e-: sky; DN: / 2.35, dark error, read noise)
    dt = float(v.read_times[-1])
    sky_e = float(v.sky[0]) * dt                                       # master sky ~ 1 (synthetic: mean 1)
    expect = sky_e / 2.35 ** 2 + (14.1 / 2.35) ** 2
    s_out = _region_stats(D, R, outside)
    for label, got in (("device", s_out["var_d"]), ("oracle", s_out["var_r"])):
        if abs(got / expect - 1.0) > 0.015:
            bad.append("%s background variance %.2f DN^2, expected ~%.2f" % (label, got, expect))
    # and the border's is the read noise alone: (14.1 / 2.35)^2 (detector.py:33, 193-198)
    s_b = _region_stats(D, R, border)
    for label, got in (("device", s_b["var_d"]), ("oracle", s_b["var_r"])):
        if abs(got / (14.1 / 2.35) ** 2 - 1.0) > 0.03:
            bad.append("%s border variance %.2f DN^2" % (label, got))
    assert not bad, "; ".join(bad)


def _fullsize_cfg4_ensemble():
    # the same comparison on the workload `bench.py` times (cfg4: 1014^2, NSAMP 16, 128 sub-samples, 10^9 electrons, every
    # detector effect on but the cosmic rays): fewer frames -- an exposure of the reference's thrower is seconds on all
    # the box's cores -- over twenty times as many pixels
    m_dev, m_ref = 48 * SCALE, 16 * SCALE
    ncpu = max(2, min(64, os.cpu_count() or 2))
    v, D, R, star = _exposure_ensembles(m_dev, m_ref, visit="cfg4", threads=(ncpu, max(1, ncpu // 2)))
    S = D.shape[1]
    interior = np.zeros((S, S), dtype=bool)
    interior[5:-5, 5:-5] = True
    inside, outside, border = interior & (star > 30.0), interior & (star < 0.5), ~interior
    assert inside.sum() > 50000 and outside.sum() > 300000
    bad = []
    for name, sel, jitter_floor in (("inside", inside, 0.01), ("outside", outside, 0.0), ("border", border, 0.0)):
        s = _region_stats(D, R, sel)
        report("exposure/cfg4/%s" % name, **dict(s, m_dev=m_dev, m_ref=m_ref))
        if abs(s["z_mean"]) > 5.0 * s["z_mean_se"] + jitter_floor:
            bad.append("%s: pixel means differ, mean z %.4f (se %.4f)" % (name, s["z_mean"], s["z_mean_se"]))
        if abs(s["z_std"] - s["z_std_expect"]) > 5.0 * s["z_std_se"] + 0.02:
            bad.append("%s: spread of z %.4f, expected %.4f" % (name, s["z_std"], s["z_std_expect"]))
        if abs(s["log_var"]) > 5.0 * s["log_var_se"] + 0.01:
            bad.append("%s: pixel variances differ, mean log ratio %.4f (se %.4f)" % (name, s["log_var"],
                                                                                    s["log_var_se"]))
    assert not bad, "; ".join(bad)


# minutes of reference C at 10^9 electrons per exposure: collected only with WAYNE_ENSEMBLE_FULLSIZE=1 (the default run holds
# the cut-down form below and reports no skip; profiles/r04/ensemble_parity.json keys exposure/cfg4/* are such a run)
if os.environ.get("WAYNE_ENSEMBLE_FULLSIZE"):
    test_production_exposures_of_the_benchmarked_configuration_against_reference_driven_oracle_ensemble = _fullsize_cfg4_ensemble


def _region_stats_small_reference(D, R, sel):
    """Per-pixel figures for a SMALL reference ensemble (4 exposures): the difference of the pixel means over the
    device ensemble's own variance -- under equal laws a Student t with M_dev - 1 degrees of freedom (the Welch form
    with its ~4.6 would have no finite kurtosis) -- and the log variance ratio with its exact small-sample bias
    (digamma) and variance (trigamma) taken out."""
    Md, Mr = D.shape[0], R.shape[0]
    md, mr = D.mean(axis=0)[sel], R.mean(axis=0)[sel]
    vd, vr = D.var(axis=0, ddof=1)[sel], R.var(axis=0, ddof=1)[sel]
    z = (md - mr) / np.sqrt(vd * (1.0 / Md + 1.0 / Mr))
    n = int(sel.sum())
    nu = Md - 1.0
    z_sd = float(np.sqrt(nu / (nu - 2.0)))
    lr = np.log(vd / vr) - es.log_s2_bias(Md) + es.log_s2_bias(Mr)
    return dict(n=n, z_mean=float(z.mean()), z_mean_se=z_sd / np.sqrt(n), z_std=float(z.std(ddof=1)), z_std_expect=z_sd,
                z_std_se=float(z_sd * np.sqrt((6.0 / (nu - 4.0) + 2.0) / (4.0 * n))),
                log_var=float(lr.mean()),
                log_var_se=float(np.sqrt(polygamma(1, (Md - 1) / 2.0) + polygamma(1, (Mr - 1) / 2.0)) / np.sqrt(n)),
                mean_d=float(md.mean()), mean_r=float(mr.mean()), var_d=float(vd.mean()), var_r=float(vr.mean()),
                z_max=float(np.abs(z).max()))


def test_benchmarked_configuration_against_a_small_reference_driven_ensemble():
    # the exposure-level comparison on the workload `bench.py` times (cfg4: 1014^2, NSAMP 16, 128
    # sub-samples, 10^9 electrons, every detector effect on but the cosmic rays) in the DEFAULT run -- cut down to 16
    # production exposures beside 4 of `ExposureOracle(thrower="ref", draws=LegacyDraws)` (the reference's C thrower on
    # the box's cores and numpy's legacy generator in the reference's call order: ~10 s each), the bands widened for the
    # small reference side as their sampling distributions say.  The 48-vs-16 run stays behind WAYNE_ENSEMBLE_FULLSIZE.
    m_dev, m_ref = 16, 4
    ncpu = max(2, min(64, os.cpu_count() or 2))
    v, D, R, star = _exposure_ensembles(m_dev, m_ref, visit="cfg4", threads=(ncpu, max(1, ncpu // 2)))
    S = D.shape[1]
    interior = np.zeros((S, S), dtype=bool)
    interior[5:-5, 5:-5] = True
    inside, outside, border = interior & (star > 30.0), interior & (star < 0.5), ~interior
    assert inside.sum() > 50000 and outside.sum() > 300000
    bad = []
    for name, sel, jitter_floor in (("inside", inside, 0.02), ("outside", outside, 0.0), ("border", border, 0.0)):
        s = _region_stats_small_reference(D, R, sel)
        report("exposure/cfg4_quick/%s" % name, **dict(s, m_dev=m_dev, m_ref=m_ref))
        if abs(s["z_mean"]) > 5.0 * s["z_mean_se"] + jitter_floor:
            bad.append("%s: pixel means differ, mean z %.4f (se %.4f)" % (name, s["z_mean"], s["z_mean_se"]))
        if abs(s["z_std"] - s["z_std_expect"]) > 5.0 * s["z_std_se"] + 0.03:
            bad.append("%s: spread of z %.4f, expected %.4f" % (name, s["z_std"], s["z_std_expect"]))
        if abs(s["log_var"]) > 5.0 * s["log_var_se"] + 0.01:
            bad.append("%s: pixel variances differ, mean log ratio %.4f (se %.4f)" % (name, s["log_var"],
                                                                                    s["log_var_se"]))
    # absolute scale of the background and of the reference pixels, as the reference's stages say (device side: 16 frames)
    dt = float(v.read_times[-1])
    expect = float(v.sky[0]) * dt / 2.35 ** 2 + (14.1 / 2.35) ** 2
    got = _region_stats_small_reference(D, R, outside)["var_d"]
    if abs(got / expect - 1.0) > 0.02:
        bad.append("device background variance %.2f DN^2, expected ~%.2f" % (got, expect))
    got_b = _region_stats_small_reference(D, R, border)["var_d"]
    if abs(got_b / (14.1 / 2.35) ** 2 - 1.0) > 0.03:
        bad.append("device border variance %.2f DN^2" % got_b)
    assert not bad, "; ".join(bad)
