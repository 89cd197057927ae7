"""CPU: the extreme-value checkers of tests/extreme_stats.py, calibrated on numpy draws of the exact laws and shown to
FAIL on the defects they exist for (negative controls) -- before tests/test_extremes_gpu.py trusts them with the device.

The injected defect is the historical one: ~500 spurious electrons in ONE pixel of a 10^6-pixel frame (VERDICT r04;
DESIGN.md "A search must be able to stop").  The moment statistic that missed it for three rounds is evaluated beside
the checker, on the same data, to show the difference.
"""
import numpy as np
import pytest
from scipy import stats

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
    win = xs.thrower_window_moments(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], 0, n, 0, n)
    np.testing.assert_allclose(win, mean, rtol=1e-12, atol=1e-14)
    sub = xs.thrower_window_moments(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], 5, 40, 3, 33)
    np.testing.assert_allclose(sub, mean[3:33, 5:40], rtol=1e-12, atol=1e-14)
    assert mean[0].sum() == 0 and mean[:, 0].sum() == 0 and abs(mean.sum() - total) < 1e-9
