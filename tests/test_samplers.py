"""Distribution tests of the oracle's counter-keyed samplers (CPU only).

The reference draws Poisson / normal variates from numpy's legacy MT19937
stream (exposure_generator.py:495,626,725; detector.py:191,198;
cosmic_rays.py:127) and throws electrons one by one (pyparallel_menu.c:87-108).
A sharded device path cannot replay that stream, so the oracle restates the
same published algorithms over Philox counters (oracle/noise_oracle.c,
oracle/split_oracle.c); these tests pin those restatements to scipy.stats.
The device is then compared with the oracle on the same counters in the
`-m gpu` tests.
"""
import numpy as np
import pytest
from scipy import stats

from oracle import clib

M = 200000
P_MIN = 1e-4      # a correct sampler fails one of these ~20 tests once in ~500 runs; the seeds are fixed


def chi2_pvalue(sample, pmf, lo, hi):
    """Chi-square p-value of integer `sample` against pmf(k), cells pooled to expectation >= 10."""
    ks = np.arange(lo, hi + 1)
    expect = pmf(ks) * sample.size
    obs = np.bincount((np.clip(sample, lo, hi) - lo).astype(np.int64), minlength=ks.size).astype(float)
    # the clipped end cells take the tails
    expect[0] += (stats_sum_below(pmf, lo)) * sample.size
    expect[-1] += max(0.0, sample.size - expect.sum())
    cells_o, cells_e, o_acc, e_acc = [], [], 0.0, 0.0
    for o, e in zip(obs, expect):
        o_acc += o
        e_acc += e
        if e_acc >= 10.0:
            cells_o.append(o_acc)
            cells_e.append(e_acc)
            o_acc = e_acc = 0.0
    if e_acc > 0 and cells_e:
        cells_o[-1] += o_acc
        cells_e[-1] += e_acc
    o, e = np.array(cells_o), np.array(cells_e)
    e *= o.sum() / e.sum()
    stat = ((o - e) ** 2 / e).sum()
    return stats.chi2.sf(stat, len(o) - 1)


def stats_sum_below(pmf, lo):
    return float(pmf(np.arange(max(lo - 2000, 0), lo)).sum()) if lo > 0 else 0.0


def support(mean, sd):
    return max(int(mean - 8 * sd) - 2, 0), int(mean + 8 * sd) + 8


@pytest.mark.parametrize("lam", [0.02, 0.7, 3.0, 9.99, 10.0, 37.5, 255.0, 1.0e4, 2.5e6])
def test_poisson_f64_counter_streams(lam):
    L = clib.lib()
    out = np.empty(M)
    L.wayne_oracle_poisson_f64(np.full(M, lam), M, 11, 1, 0, 3, 5, out)
    assert np.all(out == np.floor(out)) and out.min() >= 0
    lo, hi = support(lam, np.sqrt(lam))
    assert chi2_pvalue(out, lambda k: stats.poisson.pmf(k, lam), lo, hi) > P_MIN
    # a different counter word gives a different, equally distributed sample
    out2 = np.empty(M)
    L.wayne_oracle_poisson_f64(np.full(M, lam), M, 11, 1, 0, 4, 5, out2)
    assert not np.array_equal(out, out2)
    assert abs(out2.mean() - lam) < 5 * np.sqrt(lam / M)


@pytest.mark.parametrize("lam", [0.004, 0.02, 0.7, 2.5, 6.0, 9.99, 10.0, 37.5, 1.0e4])
def test_stellar_counts_sampler(lam):
    # the stellar counts (stage COUNTS): inversion from one uniform below a mean of 10, PTRS from 10
    L = clib.lib()
    out = np.empty(M)
    L.wayne_oracle_poisson_counts_f64(np.full(M, lam), M, 11, 1, 0, 3, 5, out)
    assert np.all(out == np.floor(out)) and out.min() >= 0
    lo, hi = support(lam, np.sqrt(lam))
    assert chi2_pvalue(out, lambda k: stats.poisson.pmf(k, lam), lo, hi) > P_MIN
    assert abs(out.mean() - lam) < 5 * np.sqrt(lam / M) and abs(out.var() - lam) < 6 * lam * np.sqrt(2.0 / M) + 6 * np.sqrt(lam / M)
    if lam >= 10.0:                                            # the PTRS branch is the generic sampler's
        ref = np.empty(M)
        L.wayne_oracle_poisson_f64(np.full(M, lam), M, 11, 1, 0, 3, 5, ref)
        np.testing.assert_array_equal(out, ref)
    else:                                                      # one word per draw: inversion is monotone in it
        blocks = np.empty((M, 4), dtype=np.uint32)
        L.wayne_oracle_philox_blocks(np.arange(M, dtype=np.uint32), M, 0, 3, 5, 11, 1, blocks)
        order = np.argsort(blocks[:, 0], kind="stable")
        assert np.all(np.diff(out[order]) >= 0)


@pytest.mark.parametrize("lam", [0.3, 9.5, 10.5, 80.0, 255.9, 256.0, 3000.0])
def test_sky_poisson_step_fp32_and_fp64_branches(lam):
    # per-pixel seeded streams, float32 sampler below 256, float64 above (k_ramp's sky draw)
    L = clib.lib()
    idx = np.arange(M, dtype=np.uint32)
    st = np.empty(M * 4, dtype=np.uint32)
    L.wayne_oracle_seed_streams(idx, M, 99, 3, 7, st)
    lamf = np.full(M, lam, dtype=np.float32)
    lo, hi = support(lam, np.sqrt(lam))
    first = np.empty(M)
    L.wayne_oracle_poisson_sky_step(lamf, M, st, first)
    second = np.empty(M)
    L.wayne_oracle_poisson_sky_step(lamf, M, st, second)       # the next read's draw of the same streams
    for s in (first, second):
        assert chi2_pvalue(s, lambda k: stats.poisson.pmf(k, float(lamf[0])), lo, hi) > P_MIN
    r = np.corrcoef(first, second)[0, 1]
    assert abs(r) < 5 / np.sqrt(M)


def test_normal_step_is_standard_normal_and_pairs_are_independent():
    L = clib.lib()
    idx = np.arange(M, dtype=np.uint32)
    st = np.empty(M * 4, dtype=np.uint32)
    L.wayne_oracle_seed_streams(idx, M, 5, 6, 0, st)
    z0, z1 = np.empty(M, np.float32), np.empty(M, np.float32)
    L.wayne_oracle_normal_step(M, st, z0, z1)
    for z in (z0, z1):
        assert stats.kstest(z.astype(float), "norm").pvalue > P_MIN
        assert abs(z.mean()) < 5 / np.sqrt(M) and abs(z.var() - 1) < 5 * np.sqrt(2.0 / M)
    assert abs(np.corrcoef(z0, z1)[0, 1]) < 5 / np.sqrt(M)
    assert abs(stats.kurtosis(z0.astype(float))) < 0.06


@pytest.mark.parametrize("n,p", [(40, 0.1), (1000, 0.004), (7, 0.5), (37, 0.97),          # inversion (BINV)
                                 (1000, 0.3), (2000, 0.7), (50000, 0.5), (400, 0.03),     # rejection (BTRS)
                                 (100000, 1e-5), (1, 0.25), (33, 0.999)])
def test_binomial_fp32(n, p):
    x = clib.binomial_vec(np.full(M, n, np.float32), np.full(M, p, np.float32), seed=5, subsample=2, exposure=9)
    assert np.all(x == np.floor(x)) and x.min() >= 0 and x.max() <= n
    pf = float(np.float32(p))
    lo, hi = support(n * pf, np.sqrt(n * pf * (1 - pf)))
    hi = min(hi, n)
    assert chi2_pvalue(x, lambda k: stats.binom.pmf(k, n, pf), lo, hi) > P_MIN


def test_binomial_degenerate_arguments():
    n = np.array([0, 10, 10, 10, 5], np.float32)
    p = np.array([0.5, 0.0, 1.0, -0.2, 1.5], np.float32)
    np.testing.assert_array_equal(clib.binomial_vec(n, p, seed=1), [0, 0, 10, 0, 5])


def _cell_probs(pos, sigma, n):
    edges = np.arange(n + 1, dtype=float)
    cdf = stats.norm.cdf((edges - pos) / sigma)
    return np.diff(cdf)


@pytest.mark.parametrize("sigma,fx,fy", [(0.55, 0.5, 0.5), (0.8, 0.07, 0.93), (0.89, 0.999, 0.001)])
def test_split_oracle_lone_bin_is_the_multinomial_of_the_narrow_gaussian(sigma, fx, fy):
    # ratio 0: every electron is narrow -> the whole frame is one multinomial draw
    N, n = 40, 3000000
    x, y = 20 + fx, 17 + fy
    f = clib.psf_split_oracle([n], [x], [y], [0.0], [sigma], [5.0], N, seed=8, exposure=1, subsample=4)
    f = f.reshape(N, N)
    assert f.sum() == n
    p = np.outer(_cell_probs(y, sigma, N), _cell_probs(x, sigma, N))
    big = p * n >= 10
    o = np.append(f[big], f[~big].sum()).astype(float)
    e = np.append(p[big], p[~big].sum()) * n
    keep = e > 0
    stat = ((o[keep] - e[keep]) ** 2 / e[keep]).sum()
    assert stats.chi2.sf(stat, keep.sum() - 1) > P_MIN
    # marginals: columns and rows are binomial chains of their own
    for marg, pm in ((f.sum(axis=0), _cell_probs(x, sigma, N)), (f.sum(axis=1), _cell_probs(y, sigma, N))):
        z = (marg - n * pm) / np.sqrt(np.maximum(n * pm * (1 - pm), 1e-9))
        assert np.abs(z[n * pm > 50]).max() < 5


def test_upper_tail_fit_against_erfc():
    # the cell masses of the multinomial are differences of this tail: 2e-7 absolute, 5e-6 relative out to the cut
    from scipy.special import erfc
    t = np.linspace(0.0, 6.5, 6501)
    got = clib.upper_tail(t)
    ref = 0.5 * erfc(t / np.sqrt(2.0))
    assert np.abs(got - ref).max() < 2.5e-7
    assert np.abs(got / ref - 1).max() < 6e-6
    assert np.all(np.diff(got) <= 0)                   # monotone: no negative cell mass
    assert np.all(clib.upper_tail([6.5001, 7.0, 30.0]) == 0.0)


@pytest.mark.parametrize("dy,dsig", [(0.001, 0.0005), (0.01, 0.004), (0.0, 0.0)])
def test_split_oracle_pooled_rows_keep_every_pixel_marginal(dy, dsig):
    # 64 neighbouring bins, all narrow: groups of 16 pool their row chains (oracle/split_oracle.c so_narrow_pooled).
    # Each pixel stays a sum over the bins of Binomial(n_b, p_b): mean AND variance against the closed form, the
    # column sums likewise (dy, dsig set how much of a bin is "residual": ~0.3 % / ~15 % / nothing).
    N, nb, reps = 96, 64, 1500
    b = np.arange(nb)
    x = 40.3 + 0.04 * b
    y = 40.3 + dy * b
    sl = 0.7 + dsig * (b % 16)
    counts = np.full(nb, 2000, dtype=np.int32)
    acc = np.zeros((N, N))
    acc2 = np.zeros((N, N))
    col = np.zeros(N)
    col2 = np.zeros(N)
    for r in range(reps):
        f = clib.psf_split_oracle(counts, x, y, np.zeros(nb), sl, np.full(nb, 5.0), N, seed=7, exposure=r, subsample=3)
        f = f.reshape(N, N).astype(float)
        assert f.sum() == counts.sum()
        acc += f
        acc2 += f * f
        c = f.sum(axis=0)
        col += c
        col2 += c * c
    mean, var = acc / reps, acc2 / reps - (acc / reps) ** 2
    e, ev, ec, ecv = np.zeros((N, N)), np.zeros((N, N)), np.zeros(N), np.zeros(N)
    for i in range(nb):
        P, Q = _cell_probs(x[i], sl[i], N), _cell_probs(y[i], sl[i], N)
        p = np.outer(Q, P)
        e += counts[i] * p
        ev += counts[i] * p * (1 - p)
        ec += counts[i] * P
        ecv += counts[i] * P * (1 - P)
    m = e > 5
    z = (mean - e)[m] / np.sqrt(ev[m] / reps)
    assert m.sum() > 30
    assert abs(z.mean()) < 4 / np.sqrt(m.sum()) and np.abs(z).max() < 4.5
    ratio = (var / np.where(m, ev, 1))[m]
    assert abs(ratio.mean() - 1) < 4 * np.sqrt(2.0 / reps / m.sum()) + 0.01
    cm = ec > 50
    cr = ((col2 / reps - (col / reps) ** 2) / np.where(cm, ecv, 1))[cm]
    assert np.abs(cr - 1).max() < 5 * np.sqrt(2.0 / reps)


def test_split_oracle_matches_per_electron_oracle_in_distribution():
    # the same spectrum thrown both ways, many seeds: equal pixel means within the Poisson error
    rng = np.random.default_rng(3)
    W, N = 60, 64
    counts = rng.integers(0, 4000, W).astype(np.int32)
    counts[::7] = rng.integers(0, 40, counts[::7].size)        # thin bins are thrown whole, one by one
    x = np.linspace(8.3, 55.1, W)
    y = 30.2 + 0.01 * (x - 8)
    ratio = np.full(W, 0.22)
    sl = np.linspace(0.5, 0.9, W)
    sh = np.full(W, 3.1)
    a = np.zeros(N * N)
    b = np.zeros(N * N)
    reps = 40
    for s in range(reps):
        fa = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=100 + s, exposure=0, subsample=0)
        fb = clib.psf_philox_oracle(counts, x, y, ratio, sl, sh, N, N, 100 + s, 0, 0)
        assert abs(int(fa.sum()) - int(fb.sum())) <= 40         # only the far wide tail leaves the frame
        a += fa
        b += fb
    big = (a + b) > 400
    z = (a[big] - b[big]) / np.sqrt(a[big] + b[big])
    assert big.sum() > 300
    assert abs(z.mean()) < 5 / np.sqrt(big.sum())
    assert 0.85 < z.std() < 1.15


def test_split_oracle_unsplit_bins_without_lanes_are_the_per_electron_thrower():
    # below split_min narrow electrons nothing is split; with the lane rule off (lane_max = 0) such bins are
    # numbered and thrown exactly as the per-electron oracle does
    W, N = 30, 48
    counts = np.full(W, 30, np.int32)
    x = np.linspace(5.5, 40.5, W)
    y = np.full(W, 20.25)
    ratio = np.full(W, 0.2)
    sl, sh = np.full(W, 0.7), np.full(W, 2.5)
    a = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=3, exposure=2, subsample=1, lane_max=0)
    b = clib.psf_philox_oracle(counts, x, y, ratio, sl, sh, N, N, 3, 2, 1)
    np.testing.assert_array_equal(a, b)
    # a PSF too wide for the +-6 px window is never split either; 5000 electrons exceed a lane's cap (4096),
    # so these bins are shared out from the block streams: the per-electron thrower again
    counts[:] = 5000
    sl[:] = 1.0
    a = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=3, exposure=2, subsample=1)
    b = clib.psf_philox_oracle(counts, x, y, ratio, sl, sh, N, N, 3, 2, 1)
    np.testing.assert_array_equal(a, b)
    # ... and at 4000 electrons each bin is thrown from its own stream: a different sample of the same distribution
    counts[:] = 4000
    a = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=3, exposure=2, subsample=1)
    b = clib.psf_philox_oracle(counts, x, y, ratio, sl, sh, N, N, 3, 2, 1)
    assert not np.array_equal(a, b) and abs(int(a.sum()) - int(b.sum())) < 6 * np.sqrt(b.sum())


def test_split_oracle_lane_bins_use_their_own_streams():
    # one-by-one electrons of a bin (here: thin bins thrown whole) come from the bin's own stream (stage LANE).
    # Same distribution as the per-electron thrower; a bin's electrons do not depend on what the other bins hold.
    rng = np.random.default_rng(5)
    W, N = 400, 64
    counts = rng.integers(0, 16, W).astype(np.int32)
    x = np.linspace(10.2, 52.7, W)
    y = 30.4 + 0.01 * (x - 8)
    ratio = np.full(W, 0.3)
    sl, sh = np.linspace(0.5, 0.9, W), np.full(W, 2.2)
    a = np.zeros(N * N)
    b = np.zeros(N * N)
    for s_ in range(60):
        fa = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=300 + s_, exposure=1, subsample=2)
        fb = clib.psf_philox_oracle(counts, x, y, ratio, sl, sh, N, N, 300 + s_, 1, 2)
        assert fa.sum() == fb.sum() == counts.sum()
        a += fa
        b += fb
    big = (a + b) > 300
    z = (a[big] - b[big]) / np.sqrt(a[big] + b[big])
    assert big.sum() > 100
    assert abs(z.mean()) < 5 / np.sqrt(big.sum()) and 0.8 < z.std() < 1.2
    # with the rule switched off the same call is the per-electron thrower, bit for bit
    off = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=3, exposure=1, subsample=2, lane_max=0)
    np.testing.assert_array_equal(off, clib.psf_philox_oracle(counts, x, y, ratio, sl, sh, N, N, 3, 1, 2))
    # locality: emptying every other bin leaves the electrons of the remaining bins where they were
    one = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed=3, exposure=1, subsample=2)
    c2 = counts.copy()
    c2[1::2] = 0
    c3 = counts.copy()
    c3[0::2] = 0
    two = clib.psf_split_oracle(c2, x, y, ratio, sl, sh, N, seed=3, exposure=1, subsample=2)
    three = clib.psf_split_oracle(c3, x, y, ratio, sl, sh, N, seed=3, exposure=1, subsample=2)
    np.testing.assert_array_equal(one, two + three)


def test_lane_electron_is_the_gaussian_including_the_refined_tail():
    # a lane-thrown electron takes ONE word: 16 bits of radius (midpoint rule in u) + the far cell h = 0 subdivided
    # from the bin's side stream, 23 bits of angle.  1e7 electrons of one wide gaussian against the analytic pixel
    # probabilities, and the tail beyond the un-refined reach sqrt(2 ln 2^17) = 4.85 sigma must be populated
    W, N, sig = 2500, 96, 3.0
    x0, y0 = 48.3, 47.7
    counts = np.full(W, 4000, np.int32)
    f = clib.psf_split_oracle(counts, np.full(W, x0), np.full(W, y0), np.ones(W), np.full(W, 1.0), np.full(W, sig), N,
                              seed=77, exposure=3, subsample=9).reshape(N, N).astype(np.float64)
    n = float(counts.sum())
    assert f.sum() == n                                               # 16 sigma to the frame's edge: nothing leaves
    edges = np.arange(N + 1)
    px = np.diff(stats.norm.cdf((edges - x0) / sig))
    py = np.diff(stats.norm.cdf((edges - y0) / sig))
    expect = n * np.outer(py, px)
    big = expect > 50
    chi2 = ((f[big] - expect[big]) ** 2 / expect[big]).sum()
    assert big.sum() > 400 and abs(chi2 - big.sum()) < 5 * np.sqrt(2.0 * big.sum()), (chi2, big.sum())
    # radial tail: pixels wholly beyond r
    yy, xx = np.mgrid[0:N, 0:N]
    near = np.hypot(np.minimum(np.abs(xx - x0), np.abs(xx + 1 - x0)), np.minimum(np.abs(yy - y0), np.abs(yy + 1 - y0)))
    for r_sig in (4.0, 4.85, 5.3):
        sel = near > r_sig * sig
        got, want = f[sel].sum(), expect[sel].sum()
        assert abs(got - want) < 5 * np.sqrt(want) + 3, (r_sig, got, want)
    assert f[near > 4.85 * sig].sum() >= 20


@pytest.mark.parametrize("lam", [0.0, 0.05, 3.3, 16.4, 55.86, 146.0])
def test_sky_alias_table_is_the_poisson_pmf(lam):
    # the table decoded analytically: P(k) = (sum over columns that keep k + columns that alias to k) / 256
    t = np.zeros(256, np.uint32)
    clib.lib().wayne_oracle_sky_alias_table(float(lam), t)
    thr = (t & 0xFFFFFF).astype(float) / 2.0 ** 24
    alias = (t >> 24).astype(int)
    pmf = np.zeros(256)
    np.add.at(pmf, np.arange(256), thr / 256)
    np.add.at(pmf, alias, (1 - thr) / 256)
    want = stats.poisson.pmf(np.arange(256), lam) if lam > 0 else np.eye(256)[0]
    assert abs(pmf.sum() - 1) < 1e-6
    assert np.abs(pmf - want).max() < 2.0 ** -23           # thresholds are rounded to 24 bits
    assert abs((pmf * np.arange(256)).sum() - lam) < 1e-4 * max(lam, 1)


@pytest.mark.parametrize("bg", [2.0, 15.2, 51.8, 120.0])
def test_sky_by_levels_and_residual_is_poisson(bg):
    # Poisson(level * bg) from the shared table + Poisson((sky - level) * bg) by inversion, per pixel
    L = clib.lib()
    rng = np.random.default_rng(4)
    levels = np.float32(0.93) + np.arange(7, dtype=np.float32) * np.float32(0.021)
    tables = np.zeros((7, 256), np.uint32)
    for l in range(7):
        L.wayne_oracle_sky_alias_table(float(np.float32(levels[l] * np.float32(bg))), tables[l])
    for sky in (0.93, 0.9415, 1.0, 1.0769):
        lvl = min(6, int((np.float32(sky) - levels[0]) / np.float32(0.021)))
        lam = np.full(M, np.float32(sky) * np.float32(bg), dtype=np.float32)
        lam_level = np.full(M, levels[lvl] * np.float32(bg), dtype=np.float32)
        st = np.empty(M * 4, dtype=np.uint32)
        L.wayne_oracle_seed_streams(np.arange(M, dtype=np.uint32), M, 77, 3, int(sky * 1000), st)
        out = np.empty(M)
        L.wayne_oracle_sky_alias_step(lam, lam_level, np.full(M, lvl, np.int32), tables.ravel(), M, st, out)
        lo, hi = support(float(lam[0]), np.sqrt(float(lam[0])))
        assert chi2_pvalue(out, lambda k: stats.poisson.pmf(k, float(lam[0])), lo, hi) > P_MIN
        second = np.empty(M)
        L.wayne_oracle_sky_alias_step(lam, lam_level, np.full(M, lvl, np.int32), tables.ravel(), M, st, second)
        assert abs(np.corrcoef(out, second)[0, 1]) < 5 / np.sqrt(M)


def test_sky_plan_of_the_exposure_oracle():
    from oracle import wayne_oracle as wo
    d = wo.PhiloxDraws(3, 1, 64)
    rng = np.random.default_rng(0)
    unit = (1 + 0.05 * rng.standard_normal((64, 64))).astype(np.float32)
    unit[3, 4] = 0.0                                            # a dead sky pixel draws nothing
    unit[9, 9] = 3.5                                            # a hot one: its remainder is drawn in pieces
    d.begin_sky(unit, [2.9 * 5.0] + [10.0 * 5.0] * 14)
    plan = d._sky_plan
    assert plan is not None and plan["L"] == 7 and plan["tables"].shape == (14, 256)
    assert plan["lvl"].min() == 0 and plan["lvl"].max() == 6
    counts = np.bincount(plan["lvl"].ravel(), minlength=7)
    assert counts.min() > 0.1 * 64 * 64 and counts.max() < 0.2 * 64 * 64        # quantile levels: equal shares
    total = np.zeros((64, 64))
    for r, bg in enumerate([2.9 * 5.0] + [10.0 * 5.0] * 14):
        total += d.sky_poisson(unit * np.float32(bg), r, unit_sky=unit, bg_count=np.float32(bg))
    assert total[3, 4] == 0
    assert abs(total[9, 9] - 3.5 * (2.9 * 5.0 + 14 * 50.0)) < 6 * np.sqrt(3.5 * 714.5)
    lam = unit.astype(float) * (2.9 * 5.0 + 14 * 50.0)
    z = (total - lam) / np.sqrt(np.maximum(lam, 1))
    assert abs(z.mean()) < 5 / 64 and 0.9 < z.std() < 1.1
    # a sky too bright for the tables falls back to the direct sampler
    d2 = wo.PhiloxDraws(3, 1, 64)
    d2.begin_sky(unit, [400.0] * 3)
    assert d2._sky_plan is None


def test_seeded_stream_pair_output():
    # SeededStream::next2: a = s0 + s3 and b = s1 + s2 of ONE state transition.  The words of a pair and of
    # consecutive pairs must behave as independent uniforms for what the path asks of them: the top bits that
    # become floats (chi^2 on 64 x 64 cells of (a, b), of (b_t, a_t+1) and of (a_t, a_t+1)), means, and no linear
    # correlation; streams seeded from neighbouring Philox counters must not track each other
    L = clib.lib()
    n = 1 << 21
    st = np.ascontiguousarray(clib.philox4x32([7, 0, 3, 1], [1963, 2]))
    w = np.empty(2 * n, dtype=np.uint32)
    L.wayne_oracle_xo_pairs(st, n, w)
    a, b = w[0::2], w[1::2]
    for u, v in ((a, b), (b[:-1], a[1:]), (a[:-1], a[1:]), (b[:-1], b[1:])):
        cells = np.bincount(((u >> 26).astype(np.int64) << 6) | (v >> 26).astype(np.int64), minlength=4096)
        stat = ((cells - u.size / 4096.0) ** 2 / (u.size / 4096.0)).sum()
        assert stats.chi2.sf(stat, 4095) > P_MIN
        assert abs(np.corrcoef(u.astype(float), v.astype(float))[0, 1]) < 5 / np.sqrt(u.size)
    for u in (a, b):
        x = u.astype(float) / 2.0 ** 32
        assert abs(x.mean() - 0.5) < 5 / np.sqrt(12.0 * n) and stats.kstest(x[:200000], "uniform").pvalue > P_MIN
        # every bit is fair
        for bit in (0, 1, 8, 16, 23, 31):
            ones = int(((u >> bit) & 1).sum())
            assert abs(ones - n / 2) < 5 * np.sqrt(n / 4)
    # 4096 streams seeded from consecutive counters, first pair of each
    first = np.empty((4096, 2), dtype=np.uint32)
    for i in range(4096):
        s_i = np.ascontiguousarray(clib.philox4x32([i, 0, 3, 1], [1963, 2]))
        L.wayne_oracle_xo_pairs(s_i, 1, first[i])
    x = first.astype(float) / 2.0 ** 32
    assert stats.kstest(x[:, 0], "uniform").pvalue > P_MIN and stats.kstest(x[:, 1], "uniform").pvalue > P_MIN
    assert abs(np.corrcoef(x[:-1, 0], x[1:, 0])[0, 1]) < 5 / 64.0
