"""CPU: the extreme-value checkers of tests/extreme_stats.py, calibrated on numpy draws of the exact laws and shown to
FAIL on the defects they exist for (negative controls) -- before tests/test_extremes_gpu.py trusts them with the device.

The injected defect is the historical one: ~500 spurious electrons in ONE pixel of a 10^6-pixel frame (VERDICT r04;
HISTORY.md "A search must be able to stop").  The moment statistic that missed it for three rounds is evaluated beside
the checker, on the same data, to show the difference.
"""
import numpy as np
import pytest
from scipy import special, stats

import extreme_stats as xs

N_PIX = 1014 * 1014


def sky_frame(rng, lam=50.0):
    lam_px = lam * (1.0 + 0.05 * np.sin(np.arange(N_PIX) * 1e-3))
    return rng.poisson(lam_px).astype(np.float64), lam_px


def test_checker_passes_on_the_exact_laws():
    rng = np.random.default_rng(1)
    bad = []
    for trial in range(3):
        k, lam = sky_frame(rng, lam=[14.0, 50.0, 0.3][trial])
        bad += xs.check(xs.poisson_tails(k, lam, rng), "poisson %d" % trial)
    x = rng.normal(3.0, 6.0, 4 * N_PIX)
    bad += xs.check(xs.normal_tails(x, 3.0, 6.0), "normal")
    lam = np.full(N_PIX, 700.0)
    y = rng.poisson(lam) / 2.35 + rng.normal(7.0, 6.0, N_PIX)
    bad += xs.check(xs.poisson_plus_normal_tails(y, lam, 2.35, 7.0, 6.0), "read")
    # (two-dimensional frames with per-pixel planes of rates, means and sigmas, as the GPU tests hand them in)
    lam2 = np.full((1014, 1014), 300.0)
    lam2[:5] = 0.0
    sig2 = np.full((1014, 1014), 6.0)
    y2 = rng.poisson(lam2) / 2.35 + rng.normal(1.0, sig2)
    bad += xs.check(xs.poisson_plus_normal_tails(y2, lam2, 2.35, np.full((1014, 1014), 1.0), sig2), "read 2-d")
    bad += xs.check(xs.normal_tails(y2[:5], np.full((5, 1014), 1.0), sig2[:5]), "normal 2-d", qs=())
    assert not bad, "; ".join(bad)


def test_tiny_rates_keep_their_tail_frequencies():
    # lam << 1: almost every draw is 0 and the randomised transform spreads those over (lam, 1) -- they are candidates
    rng = np.random.default_rng(2)
    lam = np.full(3 * N_PIX, 2e-5)
    k = rng.poisson(lam).astype(float)
    t = xs.poisson_tails(k, lam, rng)
    assert t.u_hi.size == k.size and not xs.check(t, "tiny")
    assert abs((t.u_hi < 1e-4).sum() - k.size * 1e-4) < 6 * np.sqrt(k.size * 1e-4)


def test_one_pixel_with_500_spurious_electrons_is_caught_where_the_moment_test_is_blind():
    rng = np.random.default_rng(3)
    k, lam = sky_frame(rng)
    assert not xs.check(xs.poisson_tails(k, lam, rng), "clean")
    k[123456] += 500.0                                      # the defect of rounds 1-3, in one pixel of 10^6
    bad = xs.check(xs.poisson_tails(k, lam, rng), "runaway")
    assert bad and "most extreme high draw" in bad[0]
    # ... the statistic that was in the suite all along (tests/test_configs_gpu.py: dispersion index, 6 sigma band)
    disp = ((k - lam) ** 2 / lam).sum()
    assert abs(disp - N_PIX) < 6 * np.sqrt(2.0 * N_PIX), "the moment test does not see it -- which is the point"
    # a far smaller excess is caught as well: +45 electrons on a mean of 50 (6.4 sigma of ONE pixel)
    k[123456] -= 455.0
    assert xs.check(xs.poisson_tails(k, lam, rng), "excess45")
    # ... in a read of the background (sky / gain + dark + read noise): +500 e- = +213 DN on a sigma of 13 DN
    lam_c = np.full(N_PIX, 700.0)
    y = rng.poisson(lam_c) / 2.35 + rng.normal(7.0, 6.0, N_PIX)
    assert not xs.check(xs.poisson_plus_normal_tails(y, lam_c, 2.35, 7.0, 6.0), "clean read")
    y[777] += 500.0 / 2.35
    assert xs.check(xs.poisson_plus_normal_tails(y, lam_c, 2.35, 7.0, 6.0), "runaway read")


def test_a_cut_off_tail_is_caught():
    rng = np.random.default_rng(4)
    # a sampler whose search is capped (counts above mean + 3.8 sigma come back as the cap): no single draw is
    # impossible, but the far tail is empty
    k, lam = sky_frame(rng, 50.0)
    cap = np.floor(lam + 3.8 * np.sqrt(lam))
    bad = xs.check(xs.poisson_tails(np.minimum(k, cap), lam, rng), "capped")
    assert any("beyond the 1e-05 tail" in b for b in bad), bad
    # a normal generator that never leaves 4.2 sigma (a radius word with too few bits)
    z = rng.normal(0, 1, 8 * N_PIX)
    z = z[np.abs(z) < 4.2]
    bad = xs.check(xs.normal_tails(z, 0.0, 1.0), "short normal")
    assert any("1e-05" in b or "1e-06" in b for b in bad), bad
    # and a tail that is too HEAVY at the 1e-4 level without any impossible draw (1.5 % of the pixels at a 20 % higher rate)
    k2, lam2 = sky_frame(rng, 50.0)
    hot = rng.random(N_PIX) < 0.015
    k2[hot] = rng.poisson(lam2[hot] * 1.2)
    bad = xs.check(xs.poisson_tails(k2, lam2, rng), "heavy")
    assert any("beyond the" in b for b in bad), bad


def test_bernoulli_sums_are_bounded_by_the_poisson_law():
    # a pixel's electron count from the thrower is a sum of independent Bernoullis (one per electron of every bin): the
    # Poisson law of the same mean bounds both of its tails, so the bound never raises a false alarm ...
    rng = np.random.default_rng(5)
    n, p = 4000, rng.uniform(0.0, 0.3, N_PIX // 4)
    k = rng.binomial(n, p).astype(float)
    t = xs.bernoulli_sum_tails(k, n * p, rng)
    assert not xs.check(t, "binomial under the bound", exact_frequencies=False)
    # ... and still catches the spurious 500
    k[1000] += 500
    assert xs.check(xs.bernoulli_sum_tails(k, n * p, rng), "runaway", exact_frequencies=False)


def test_window_moments_agree_with_the_whole_frame_law():
    import ensemble_stats as es
    from conftest import load_golden_psf
    g = load_golden_psf("s64_t3")
    n = g["nr"]
    mean, _, _, total = es.analytic_moments(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], n)
    _, var, _, _ = es.analytic_moments(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], n)
    win, second = xs.thrower_window_moments(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], 0, n, 0, n)
    np.testing.assert_allclose(win, mean, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(win - second, var, rtol=1e-10, atol=1e-14)
    sub, _ = xs.thrower_window_moments(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], 5, 40, 3, 33)
    np.testing.assert_allclose(sub, mean[3:33, 5:40], rtol=1e-12, atol=1e-14)
    assert mean[0].sum() == 0 and mean[:, 0].sum() == 0 and abs(mean.sum() - total) < 1e-9
    # the binomials of ONE pixel add up to that pixel's moments
    Y, X = np.unravel_index(np.argmax(mean), mean.shape)
    nn, pp = xs.thrower_pixel_terms(g["counts"][None, :], g["x"][None, :], g["y"][None, :], g["ratio"], g["sl"], g["sh"], X, Y)
    assert abs((nn * pp).sum() - mean[Y, X]) < 1e-9 * mean[Y, X] and abs((nn * pp * (1 - pp)).sum() - var[Y, X]) < 1e-9 * var[Y, X]


def test_saddlepoint_tails_of_a_sum_of_binomials_against_the_exact_pmf():
    # a pixel's electron count is a sum of thousands of binomials (one per bin, component and sub-sample): its exact
    # tails by the lattice saddlepoint formula, against the pmf from the FFT of the characteristic function and, for a
    # single binomial, against scipy -- over means of 20 ... 5000 electrons and 3 ... 6 sigma on both sides
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(24):
        m = int(rng.integers(50, 3000))
        n = rng.integers(1, 2000, m).astype(float)
        p = np.concatenate([rng.uniform(0, 0.4, m // 3), 10 ** rng.uniform(-8, -2, m - m // 3)])
        if trial % 3 == 0:
            n = n // 50 + 1
        p = p * min(1.0, 10 ** rng.uniform(1.3, 3.7) / (n * p).sum())
        pmf = xs.pb_pmf_fft(n, p)
        mean, sd = (n * p).sum(), np.sqrt((n * p * (1 - p)).sum())
        ge, le = np.cumsum(pmf[::-1])[::-1], np.cumsum(pmf)
        for zz in (3.0, 4.0, 5.0, 6.0):
            k = int(round(mean + zz * sd))
            if ge[k] > 1e-10:
                worst = max(worst, abs(xs.pb_tail_saddle(k, n, p, True) / ge[k] - 1.0))
            k = int(round(mean - zz * sd))
            if k >= 0 and le[k] > 1e-10:
                worst = max(worst, abs(xs.pb_tail_saddle(k, n, p, False) / le[k] - 1.0))
    assert worst < 0.08, worst          # (the worst cases are lower tails of means near 20, a few counts from zero)
    for N, P, k in [(5000, 0.3, 1650), (5000, 0.3, 1350), (200, 0.02, 15), (100000, 0.001, 150), (300, 0.1, 0), (300, 0.1, 1)]:
        if k > N * P:
            assert abs(xs.pb_tail_saddle(k, [N], [P], True) / stats.binom.sf(k - 1, N, P) - 1) < 2e-3
        else:
            assert abs(xs.pb_tail_saddle(k, [N], [P], False) / stats.binom.cdf(k, N, P) - 1) < 2e-3
    # and as a test of a SAMPLER: binomial sums drawn by numpy pass; the same sums with 6 % extra spread do not
    n = np.full(100, 160.0)
    p = rng.uniform(0.01, 0.3, 100)
    mean, sd = (n * p).sum(), np.sqrt((n * p * (1 - p)).sum())
    S = rng.binomial(n.astype(int)[None, :], p[None, :], size=(200000, 100)).sum(axis=1).astype(float)
    t = xs.poisson_binomial_tails(S, (S - mean) / sd, lambda i: (n, p), rng)
    assert not xs.check(t, "binomial sums")
    S2 = np.rint(mean + (S - mean) * 1.06)
    t2 = xs.poisson_binomial_tails(S2, (S2 - mean) / sd, lambda i: (n, p), rng)
    assert xs.check(t2, "6 % too wide")


def _ptrs(lam, n, us_min, rng):
    """Hoermann's PTRS in numpy (the algorithm of wayne_amd/csrc/samplers.h PtrsSetup), with the quick-acceptance region
    us >= us_min (0.07 in the algorithm)."""
    slam, loglam = np.sqrt(lam), np.log(lam)
    b = 0.931 + 2.53 * slam
    a = -0.059 + 0.02483 * b
    invalpha = 1.1239 + 1.1328 / (b - 3.4)
    vr = 0.9277 - 3.6224 / (b - 2)
    out = np.empty(0)
    while out.size < n:
        m = int((n - out.size) * 1.3) + 1000
        U, V = rng.random(m) - 0.5, rng.random(m)
        us = 0.5 - np.abs(U)
        k = np.floor((2 * a / us + b) * U + lam + 0.43)
        quick = (us >= us_min) & (V <= vr)
        undecided = ~quick & ~((k < 0) | ((us < 0.013) & (V > us)))
        acc = quick.copy()
        kk = k[undecided]
        acc[undecided] = (np.log(V[undecided]) + np.log(invalpha) - np.log(a / us[undecided] ** 2 + b)
                          <= -lam + kk * loglam - special.gammaln(kk + 1))
        out = np.concatenate([out, k[acc]])
    return out[:n]


def test_the_body_of_a_law_is_out_of_the_tails_sight_and_in_the_pit_tests():
    # The audit's mutant (scripts/mutation_audit.py ptrs_quick_accept): PTRS accepting without its density test for
    # 0.03 <= us < 0.07 puts mass between 1.9 and 3.2 sigma -- + 8 % of variance at a rate of 60 -- and leaves the law from
    # 3.7 sigma on alone.  The tail checks pass on it; the randomised-PIT chi-square does not.  And both pass on the algorithm
    # as published, and on numpy's normals / fail on normals 1 % too wide.
    rng = np.random.default_rng(17)
    lam = np.full(2000000, 60.0)
    good, bad = _ptrs(60.0, lam.size, 0.07, rng), _ptrs(60.0, lam.size, 0.03, rng)
    assert abs(bad.var() / 60.0 - 1.08) < 0.02 and abs(good.var() / 60.0 - 1.0) < 0.005
    for k in (good, bad):
        assert not xs.check(xs.poisson_tails(k, lam, rng), "PTRS", qs=(1e-4, 1e-5))      # tails: blind to it
    chi2_good, p_good, n = xs.poisson_pit_uniformity(good, lam, rng)
    chi2_bad, p_bad, _ = xs.poisson_pit_uniformity(bad, lam, rng)
    assert n == lam.size and p_good > 1e-3 and p_bad < 1e-12 and chi2_bad > 100 * chi2_good
    x = rng.normal(3.0, 2.0, 3000000)
    assert xs.normal_pit_uniformity(x, 3.0, 2.0, rng)[1] > 1e-3
    assert xs.normal_pit_uniformity(x, 3.0, 2.02, rng)[1] < 1e-9
    # a subset is drawn when there are more draws than max_draws
    assert xs.normal_pit_uniformity(x, 3.0, 2.0, rng, max_draws=100000)[2] == 100000
