"""Extreme-value checks for the stochastic stages: the LARGEST single-pixel deviation against the exact law's tail.

Why: for three rounds a sky draw whose uniform fell into the rounding residue of its float32 cdf walked
512 steps -- ~500 spurious electrons in ONE pixel per exposure.  Same-counter parity could not see it (the oracle shared
the flaw) and moment tests could not either: a dispersion index over 10^6 pixels gains 5e3 from such a pixel, its 6 sigma
band is 8.5e3.  A sum over pixels is blind to one pixel; a maximum is not.

Every check turns a draw x_i with exactly known law F_i into its tail probabilities
    u_hi = P(X > x_i) + V P(X = x_i),      u_lo = P(X < x_i) + (1 - V) P(X = x_i),        V ~ U(0, 1)
(the randomised probability integral transform: exactly uniform on (0, 1) under the law, for discrete laws too) and asks

  * family-wise: min_i u >= alpha / n -- the largest deviation of n draws is one the law produces (probability of a
    false alarm <= alpha per tail, whatever the dependence between draws);
  * tail frequencies: #(u < q) ~ Binomial(n, q) for q = 1e-4, 1e-5, 1e-6 -- the tail is neither too heavy nor CUT OFF
    (a capped search, a table that ends too early, a normal generator that never leaves 5 sigma).

Only candidates (|z| beyond a prefilter) need their exact tail; all others provably have u > the largest q.
Laws: Poisson (sky: exposure_generator.py:488-495; stellar counts: :625-628), normal (dark current, read noise:
detector.py:185-198), Poisson / gain + normal (a whole read of the background), and -- as a conservative bound -- a sum
of independent Bernoullis (a pixel's electron count from the thrower, pyparallel_menu.c:87-108) bounded by the Poisson
law of the same mean (Hoeffding 1956; Anderson & Samuels 1967: both tails of the Poisson law are the heavier).
"""
import numpy as np
from scipy import special, stats

QS = (1e-4, 1e-5, 1e-6)


class Tails(object):
    """u_hi / u_lo of the candidates of n draws (all other draws have u above `floor`)."""

    def __init__(self, n, u_hi, u_lo, floor, where_hi=None, where_lo=None):
        self.n, self.u_hi, self.u_lo, self.floor = int(n), np.asarray(u_hi), np.asarray(u_lo), float(floor)
        self.where_hi, self.where_lo = where_hi, where_lo

    def merged(self, other):
        return Tails(self.n + other.n, np.concatenate([self.u_hi, other.u_hi]), np.concatenate([self.u_lo, other.u_lo]),
                     min(self.floor, other.floor))


def poisson_tails(k, lam, rng, z_pre=2.5):
    """Randomised tail probabilities of counts k under Poisson(lam), elementwise.  Entries with lam <= 0 must hold 0."""
    k = np.asarray(k, dtype=np.float64).ravel()
    lam = np.asarray(lam, dtype=np.float64).ravel()
    assert k.shape == lam.shape
    dead = ~(lam > 0)
    if dead.any():
        assert not k[dead].any(), "counts where the rate is zero"
    live = ~dead
    z = np.zeros_like(k)
    z[live] = (k[live] - lam[live]) / np.sqrt(lam[live])
    # candidates: z beyond the prefilter -- and EVERY draw at a rate so small that z says little (at lam << 1 the
    # randomised u of a zero count is spread over (lam, 1): those draws carry the tail frequencies)
    hi = live & ((z > z_pre) | (lam < 4.0))
    lo = live & ((z < -z_pre) | (lam < 4.0))
    v = rng.random(k.size)
    u_hi = stats.poisson.sf(k[hi], lam[hi]) + v[hi] * stats.poisson.pmf(k[hi], lam[hi])
    u_lo = stats.poisson.cdf(k[lo] - 1, lam[lo]) + (1.0 - v[lo]) * stats.poisson.pmf(k[lo], lam[lo])
    # non-candidates (lam >= 4, |z| <= 2.5): P(K > k) >= 3e-3 (the Poisson upper tail is heavier than the normal one,
    # 6e-3 there) and P(K < k) >= 1.5e-3 (lam = 10: k = 3, cdf(2) = 2.8e-3; lam = 100: cdf(74) = 3.7e-3; lam = 4..6: no
    # count lies below -2.5 sigma at all): every u below 1e-3 is among the candidates
    floor = 1e-3
    return Tails(int(live.sum()), u_hi, u_lo, floor, np.nonzero(hi)[0], np.nonzero(lo)[0])


def poisson_pit_uniformity(k, lam, rng, bins=50, max_draws=4000000):
    """The BULK of a Poisson law, not its tails: the randomised probability integral transform u = P(K < k) + V P(K = k)
    of every draw (a random subset of `max_draws` when there are more) is exactly uniform on (0, 1) under the law, whatever
    the rates; a chi-square over `bins` equal bins sees a distortion of a fraction of a per cent anywhere in the pmf --
    a transformed-rejection sampler that skips its density test in part of the proposal region, a threshold table a bit
    off -- which neither the tail frequencies nor a variance within a few 1e-3 would.  Returns (chi2, p-value, n)."""
    k = np.asarray(k, dtype=np.float64).ravel()
    lam = np.asarray(lam, dtype=np.float64).ravel()
    live = lam > 0
    k, lam = k[live], lam[live]
    if k.size > max_draws:
        pick = rng.choice(k.size, max_draws, replace=False)
        k, lam = k[pick], lam[pick]
    u = stats.poisson.cdf(k - 1, lam) + rng.random(k.size) * stats.poisson.pmf(k, lam)
    hist = np.histogram(np.clip(u, 0.0, np.nextafter(1.0, 0.0)), bins=bins, range=(0.0, 1.0))[0]
    expect = k.size / float(bins)
    chi2 = float(((hist - expect) ** 2 / expect).sum())
    return chi2, float(stats.chi2.sf(chi2, bins - 1)), int(k.size)


def normal_pit_uniformity(x, mean, sigma, rng, bins=50, max_draws=4000000):
    """The bulk of a normal law: u = Phi((x - mean) / sigma) of (a random subset of) the draws is uniform on (0, 1); a
    chi-square over `bins` equal bins.  Returns (chi2, p-value, n)."""
    z = ((np.asarray(x, dtype=np.float64) - mean) / sigma).ravel()
    if z.size > max_draws:
        z = z[rng.choice(z.size, max_draws, replace=False)]
    u = special.ndtr(z)
    hist = np.histogram(np.clip(u, 0.0, np.nextafter(1.0, 0.0)), bins=bins, range=(0.0, 1.0))[0]
    expect = z.size / float(bins)
    chi2 = float(((hist - expect) ** 2 / expect).sum())
    return chi2, float(stats.chi2.sf(chi2, bins - 1)), int(z.size)


def normal_tails(x, mean, sigma, z_pre=3.0):
    x = np.asarray(x, dtype=np.float64)
    z = ((x - np.asarray(mean, dtype=np.float64)) / np.asarray(sigma, dtype=np.float64)).ravel()
    hi, lo = z > z_pre, z < -z_pre
    return Tails(z.size, special.ndtr(-z[hi]), special.ndtr(z[lo]), float(stats.norm.sf(z_pre)), np.nonzero(hi)[0],
                 np.nonzero(lo)[0])


def poisson_plus_normal_tails(x, lam, gain, mean, sigma, z_pre=3.0):
    """x = Poisson(lam) / gain + N(mean, sigma): exact tails by convolution, for the candidates."""
    x = np.asarray(x, dtype=np.float64)
    lam = np.broadcast_to(np.asarray(lam, dtype=np.float64), x.shape).ravel()
    mean = np.broadcast_to(np.asarray(mean, dtype=np.float64), x.shape).ravel()
    sigma = np.broadcast_to(np.asarray(sigma, dtype=np.float64), x.shape).ravel()
    x = x.ravel()
    mu = lam / gain + mean
    sd = np.sqrt(lam / gain ** 2 + sigma ** 2)
    z = (x - mu) / sd
    out = []
    for sel, upper in ((z > z_pre, True), (z < -z_pre, False)):
        idx = np.nonzero(sel)[0]
        u = np.empty(idx.size)
        # group candidates by (rounded) rate: the Poisson support is shared within a group
        for lo_ in range(0, idx.size, 4096):
            j = idx[lo_:lo_ + 4096]
            l_max = lam[j].max()
            kk = np.arange(0, int(l_max + 14 * np.sqrt(l_max + 1) + 30))
            pm = stats.poisson.pmf(kk[None, :], lam[j][:, None])
            t = (x[j][:, None] - kk[None, :] / gain - mean[j][:, None]) / sigma[j][:, None]
            tail = special.ndtr(-t) if upper else special.ndtr(t)
            u[lo_:lo_ + 4096] = (pm * tail).sum(axis=1)
        out.append((u, idx))
    return Tails(x.size, out[0][0], out[1][0], float(stats.norm.sf(z_pre)) * 0.3, out[0][1], out[1][1])


def bernoulli_sum_tails(k, mean, rng, z_pre=2.5):
    """A count that is a sum of independent Bernoullis with total mean `mean` (a pixel's electrons from the thrower):
    tails under Poisson(mean), which bound the true ones from above in both directions beyond mean +- 1 -- so a draw
    the bound finds impossible IS impossible, while tail frequencies may only fall short (checked one-sidedly)."""
    return poisson_tails(k, mean, rng, z_pre)


def check(t, label, alpha=1e-3, qs=QS, exact_frequencies=True, dependence=1.0):
    """-> list of failure statements (empty = pass) for one stage's Tails."""
    bad = []
    n = t.n
    bound = alpha / max(n, 1)
    for side, u in (("high", t.u_hi), ("low", t.u_lo)):
        if u.size and u.min() < bound:
            bad.append("%s: the most extreme %s draw of %d has tail probability %.2e (family-wise bound %.2e)" % (
                label, side, n, u.min(), bound))
        for q in qs:
            if q >= t.floor:
                continue
            x = int((u < q).sum())
            e = n * q
            if e < 3:
                continue
            if exact_frequencies:
                # two-sided binomial, widened for draws that are not independent (`dependence` = variance inflation)
                sd = np.sqrt(e * dependence)
                p = 2 * min(stats.norm.sf((x - e) / sd), stats.norm.cdf((x - e) / sd)) if dependence != 1.0 else \
                    stats.binomtest(x, n, q).pvalue
                if p < 1e-4:
                    bad.append("%s: %d %s draws beyond the %.0e tail, expected %.1f" % (label, x, side, q, e))
            elif x > e + 5 * np.sqrt(e) + 2:
                bad.append("%s: %d %s draws beyond the (bounding) %.0e tail, at most %.1f expected" % (label, x, side, q, e))
    return bad


def summary(t):
    out = {"n": t.n, "min_u_hi": float(t.u_hi.min()) if t.u_hi.size else None,
           "min_u_lo": float(t.u_lo.min()) if t.u_lo.size else None, "family_bound": 1e-3 / max(t.n, 1)}
    for q in QS:
        out["hi<%.0e" % q] = int((t.u_hi < q).sum())
        out["lo<%.0e" % q] = int((t.u_lo < q).sum())
        out["expect<%.0e" % q] = t.n * q
    return out


def thrower_window_moments(counts, x, y, ratio, sl, sh, x0, x1, y0, y1):
    """Exact mean and sum of squared cell probabilities of every pixel of the window [y0, y1) x [x0, x1) (frame
    coordinates) of the reference thrower's frame for FIXED counts (the law of tests/ensemble_stats.analytic_moments, on
    a window instead of the whole frame: pyparallel_menu.c:87-108).  -> (mean, second): variance = mean - second."""
    counts = np.asarray(counts, dtype=np.float64)
    n_wide = np.trunc(counts * ratio)
    n_narrow = counts - n_wide
    ex = np.arange(x0, x1 + 1, dtype=np.float64)
    ey = np.arange(y0, y1 + 1, dtype=np.float64)

    def axis(pos, sig, edges, first):
        cdf = special.ndtr((edges[None, :] - pos[:, None]) / sig[:, None])
        p = np.diff(cdf, axis=1)
        if first <= 0:                       # row / column 0 (and anything negative) never receives an electron
            p[:, :1 - first] = 0.0
        return p

    pyh, pxh, pyl, pxl = axis(y, sh, ey, y0), axis(x, sh, ex, x0), axis(y, sl, ey, y0), axis(x, sl, ex, x0)
    mean = (pyh * n_wide[:, None]).T @ pxh + (pyl * n_narrow[:, None]).T @ pxl
    second = ((pyh ** 2) * n_wide[:, None]).T @ (pxh ** 2) + ((pyl ** 2) * n_narrow[:, None]).T @ (pxl ** 2)
    return mean, second


def thrower_pixel_terms(counts, x, y, ratio, sl, sh, X, Y, p_min=1e-13):
    """The binomials whose sum is the count of pixel (Y, X) (frame coordinates, both >= 1): for every bin of every
    sub-sample handed in (counts, x, y: [sub-samples][bins]) the wide component Binomial(N_b, P_h) and the narrow one
    Binomial(n_b - N_b, P_l), N_b = (int)(n_b ratio_b) (pyparallel_menu.c:89-107).  -> (n[], p[]) with p > p_min."""
    counts = np.asarray(counts, dtype=np.float64)
    n_wide = np.trunc(counts * ratio[None, :])
    n_narrow = counts - n_wide

    def cell(pos, sig, c):
        return special.ndtr((c + 1.0 - pos) / sig[None, :]) - special.ndtr((c - pos) / sig[None, :])

    ph = cell(x, sh, float(X)) * cell(y, sh, float(Y))
    pl = cell(x, sl, float(X)) * cell(y, sl, float(Y))
    n = np.concatenate([n_wide.ravel(), n_narrow.ravel()])
    p = np.concatenate([ph.ravel(), pl.ravel()])
    keep = (p > p_min) & (n > 0)
    return n[keep], p[keep]


def pb_pmf_fft(n, p):
    """Exact pmf of S = sum_i Binomial(n_i, p_i) on 0 .. M-1 by the FFT of its characteristic function (M holds the mean
    + 14 sigma).  The yardstick of pb_tail_saddle in tests/test_extremes_cpu.py -- too slow for a million pixels."""
    n = np.asarray(n, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    mean = (n * p).sum()
    sd = np.sqrt((n * p * (1 - p)).sum())
    M = 1 << int(np.ceil(np.log2(max(64.0, mean + 14 * sd + 64))))
    e = np.exp(2j * np.pi * np.arange(M) / M) - 1
    logphi = np.zeros(M, dtype=complex)
    for i0 in range(0, n.size, 1024):
        logphi += (n[i0:i0 + 1024, None] * np.log1p(p[i0:i0 + 1024, None] * e[None, :])).sum(axis=0)
    return np.maximum(np.real(np.fft.fft(np.exp(logphi))) / M, 0.0)


def pb_tail_saddle(k, n, p, upper=True):
    """P(S >= k) (upper) or P(S <= k) (lower) of S = sum_i Binomial(n_i, p_i): the Lugannani-Rice saddlepoint formula
    with Daniels' lattice correction (u = (1 - e^-s) sqrt(K''(s))), applied to S or to -S; exact at the edge of the
    support (k <= 1 from below).  Relative error ~1e-3 (upper tails) .. 6e-2 (lower tails a few counts from zero) beyond 3 sigma for means of 20 .. 5000 (against
    pb_pmf_fft: tests/test_extremes_cpu.py) -- ample for tail probabilities compared with bounds decades away."""
    n = np.asarray(n, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    mean = (n * p).sum()
    var = (n * p * (1 - p)).sum()
    if upper and k <= 0:
        return 1.0
    if not upper:
        if k < 0:
            return 0.0
        if k <= 1:
            p0 = np.exp((n * np.log1p(-p)).sum())
            return float(p0 if k == 0 else p0 * (1.0 + (n * p / (1 - p)).sum()))
    sgn = 1.0 if upper else -1.0
    kk = sgn * k
    if sgn * mean >= kk:                 # not in this tail at all: the normal value does (callers only ask beyond 3 sigma)
        return float(stats.norm.sf((kk - sgn * mean - 0.5) / np.sqrt(var)))

    def K012(s):
        es = np.exp(sgn * s)
        d = 1 - p + p * es
        q = p * es / d
        return (n * np.log(d)).sum(), sgn * (n * q).sum(), (n * q * (1 - q)).sum()

    s = (kk - sgn * mean) / var
    for _ in range(60):
        K0, K1, K2 = K012(s)
        ds = (kk - K1) / K2
        s += ds
        if abs(ds) < 1e-12 * max(1.0, abs(s)):
            break
    K0, K1, K2 = K012(s)
    w = np.sqrt(max(2 * (s * kk - K0), 0.0))
    u = (1 - np.exp(-s)) * np.sqrt(K2)
    return float(stats.norm.sf(w) + stats.norm.pdf(w) * (1 / u - 1 / w))


def poisson_binomial_tails(k, z, terms_of, rng, z_pre=3.0):
    """Randomised tail probabilities of the counts k[i] whose law is a sum of binomials: `terms_of(i)` -> (n[], p[]);
    z[i] = (k - mean) / sigma of that law selects the candidates.  -> Tails over all of k."""
    k = np.asarray(k, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    hi, lo = np.nonzero(z > z_pre)[0], np.nonzero(z < -z_pre)[0]
    v = rng.random(k.size)
    u_hi, u_lo = np.empty(hi.size), np.empty(lo.size)
    for j, i in enumerate(hi):
        n, p = terms_of(i)
        a, b = pb_tail_saddle(k[i] + 1, n, p, True), pb_tail_saddle(k[i], n, p, True)      # P(S > k), P(S >= k)
        u_hi[j] = a + v[i] * max(b - a, 0.0)
    for j, i in enumerate(lo):
        n, p = terms_of(i)
        a, b = pb_tail_saddle(k[i] - 1, n, p, False), pb_tail_saddle(k[i], n, p, False)     # P(S < k), P(S <= k)
        u_lo[j] = a + (1.0 - v[i]) * max(b - a, 0.0)
    return Tails(k.size, u_hi, u_lo, float(stats.norm.sf(z_pre)) * 0.5, hi, lo)
