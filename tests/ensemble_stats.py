"""Two-sample statistics for ENSEMBLES of thrower frames (the production RNG modes are
"validated statistically ... per-pixel mean/variance, chi-square over ensembles").

A[m], B[m]: M_a and M_b independent integer frames of the same input from two throwers -- one of them the reference's
compiled C (`oracle/_ref`, wayne/pyparallel_menu.c:87-108, different `test` seeds).  Every figure below has a known
distribution when the two throwers draw from the same law; each test states its band in standard errors of that
distribution (5 sigma unless said otherwise), so a band is not a fitted number.

What each figure would catch (HISTORY.md section 6, "ensemble parity"):
  z_mean / z_std        a shift of any pixel's expected count (wrong cell masses, a biased sampler, a lost tail) and
                        over- or under-dispersion of the pixel counts against the reference
  log_var_ratio / fano  the variance law itself: with `N = (int)(counts*ratio)` fixed (pyparallel_menu.c:89) a pixel's
                        count is a sum of two binomials per bin, variance < mean (Fano ~ 1 - p_pixel, 0.65-0.75 in the
                        core of the trace); a thrower that drew the sigma split per electron, or per-pixel Poisson
                        counts, has a larger variance by n*ratio*(1-ratio)*(p_h - p_l)^2 (+7 to +10 % there)
  wing_chi2             the radial law of the wide gaussian far from the trace (rows 13-30 from it: 2.3-5.4 sigma_h),
                        where the 16-bit radius word, its refinement beyond 4.7 sigma and the hardware log2 / sqrt act
  total_z               electrons lost off the frame's edge (the `0 < pos < n` test, pyparallel_menu.c:92)
  row0 / col0           pixel row 0 and column 0 never receive an electron (C truncation toward zero, :91-93)
"""
import numpy as np


def trace_row_offsets(x, y, n):
    """For every pixel (row, col) of an n x n frame: row - floor(y of the trace at that column), the trace taken
    from the bin positions (linear in x over the bins, extrapolated at both ends)."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    ok = np.isfinite(x) & np.isfinite(y)
    order = np.argsort(x[ok])
    xs, ys = x[ok][order], y[ok][order]
    cols = np.arange(n) + 0.5
    if xs.size >= 2 and xs[-1] > xs[0]:
        slope = (ys[-1] - ys[0]) / (xs[-1] - xs[0])
        yt = np.interp(cols, xs, ys)
        yt = np.where(cols < xs[0], ys[0] + slope * (cols - xs[0]), yt)
        yt = np.where(cols > xs[-1], ys[-1] + slope * (cols - xs[-1]), yt)
    else:
        yt = np.full(n, ys.mean() if ys.size else 0.0)
    rows = np.arange(n)[:, None]
    return rows - np.floor(yt)[None, :]


def log_s2_bias(M):
    """E[log s^2] - log sigma^2 of a gaussian sample of M: digamma(nu / 2) - log(nu / 2), nu = M - 1 (= -1/nu - 1/(3 nu^2)
    - ...; the second term is 1.5e-3 at M = 16 -- four standard errors of a mean over 10^6 pixels)."""
    from scipy.special import digamma
    nu = M - 1.0
    return float(digamma(nu / 2.0) - np.log(nu / 2.0))


def compare(A, B, x=None, y=None, min_sum=400, min_mean_var=8.0, wing=(13, 30)):
    """-> dict of the figures named in the module docstring, each with its standard error where it has one."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    Ma, Mb = A.shape[0], B.shape[0]
    n = A.shape[1]
    sa, sb = A.sum(axis=0), B.sum(axis=0)
    ma, mb = sa / Ma, sb / Mb
    va, vb = A.var(axis=0, ddof=1), B.var(axis=0, ddof=1)
    # pixels with at least min_sum electrons summed over EACH ensemble (counts there are near-gaussian)
    bright = (sa >= min_sum) & (sb >= min_sum) & (va > 0) & (vb > 0)
    nb = int(bright.sum())
    out = {"n_bright": nb, "Ma": Ma, "Mb": Mb}
    if nb:
        z = (ma[bright] - mb[bright]) / np.sqrt(va[bright] / Ma + vb[bright] / Mb)
        # Welch's statistic: for equal variances its degrees of freedom (Welch-Satterthwaite) are
        # (1/Ma + 1/Mb)^2 / (1/(Ma^2 (Ma-1)) + 1/(Mb^2 (Mb-1))); std = sqrt(nu / (nu - 2))
        nu = (1.0 / Ma + 1.0 / Mb) ** 2 / (1.0 / (Ma * Ma * (Ma - 1.0)) + 1.0 / (Mb * Mb * (Mb - 1.0)))
        out["z_mean"], out["z_mean_se"] = float(z.mean()), 1.0 / np.sqrt(nb)
        out["z_std"], out["z_std_expect"] = float(z.std(ddof=1)), float(np.sqrt(nu / (nu - 2.0)))
        out["z_std_se"] = float(out["z_std_expect"] / np.sqrt(2.0 * nb))
        out["z_max"] = float(np.abs(z).max())
        # variance figures over the pixels with at least `min_mean_var` electrons per frame (near-gaussian counts).
        # log of a ratio of two sample variances: E[log s^2 / sigma^2] = -1/(M-1), so the mean is 1/(Mb-1) - 1/(Ma-1)
        # (taken out below); a count of mean lambda has excess kurtosis ~ 1/lambda: var(log s^2) = 2/(M-1) + 1/(lambda M)
        bv = bright & (ma >= min_mean_var) & (mb >= min_mean_var)
        out["n_var"] = int(bv.sum())
        if bv.any():
            lr = np.log(va[bv] / vb[bv])
            lam = 0.5 * (ma[bv] + mb[bv])
            se_px2 = 2.0 / (Ma - 1) + 2.0 / (Mb - 1) + (1.0 / Ma + 1.0 / Mb) / lam
            out["log_var_ratio"] = float(lr.mean() - log_s2_bias(Ma) + log_s2_bias(Mb))
            out["log_var_ratio_se"] = float(np.sqrt(se_px2.sum()) / bv.sum())
            # Fano factor of the ten per cent brightest of those pixels (the core of the trace)
            core = bv & (sa + sb >= np.percentile((sa + sb)[bv], 90))
            fa, fb = va[core] / ma[core], vb[core] / mb[core]
            out["fano_a"], out["fano_b"] = float(fa.mean()), float(fb.mean())
            out["fano_se"] = float(np.sqrt((2.0 / (Ma - 1) + 2.0 / (Mb - 1))) * 0.5 * (fa.mean() + fb.mean())
                                   / np.sqrt(core.sum()))
            out["n_core"] = int(core.sum())
    # electrons kept on the frame, per frame
    ta, tb = A.reshape(Ma, -1).sum(axis=1), B.reshape(Mb, -1).sum(axis=1)
    se = np.sqrt(ta.var(ddof=1) / Ma + tb.var(ddof=1) / Mb)
    out["total_a"], out["total_b"], out["total_se"] = float(ta.mean()), float(tb.mean()), float(se)
    out["row0"] = float(A[:, 0, :].sum() + B[:, 0, :].sum())
    out["col0"] = float(A[:, :, 0].sum() + B[:, :, 0].sum())
    if x is not None:
        off = trace_row_offsets(x, y, n)
        lo, hi = wing
        cells_a, cells_b = [], []
        # cells: (row offset, third of the frame's columns) on both sides of the trace
        thirds = np.array_split(np.arange(n), 3)
        for d in list(range(-hi, -lo + 1)) + list(range(lo, hi + 1)):
            sel = off == d
            for cols in thirds:
                m = np.zeros_like(sel)
                m[:, cols] = sel[:, cols]
                cells_a.append(sa[m].sum())
                cells_b.append(sb[m].sum())
        ca, cb = np.array(cells_a), np.array(cells_b)
        ok = (ca + cb) >= 50
        # two-sample chi-square of counts (different ensemble sizes allowed)
        ka, kb = np.sqrt(Mb / Ma), np.sqrt(Ma / Mb)
        chi2 = (((ka * ca[ok] - kb * cb[ok]) ** 2) / (ca[ok] + cb[ok])).sum()
        out["wing_chi2"], out["wing_dof"] = float(chi2), int(ok.sum())
        out["wing_electrons"] = float(ca.sum() + cb.sum())
    return out


def check(s, sigma=5.0, require_subpoisson=None):
    """Assert the bands; returns the list of failed statements (empty = pass) so a test can report all of them."""
    bad = []
    if s["n_bright"]:
        if abs(s["z_mean"]) > sigma * s["z_mean_se"]:
            bad.append("pixel means differ: mean z = %.4f (se %.4f)" % (s["z_mean"], s["z_mean_se"]))
        if abs(s["z_std"] - s["z_std_expect"]) > sigma * s["z_std_se"] + 0.01:
            bad.append("spread of pixel z: %.4f, expected %.4f (se %.4f)" % (s["z_std"], s["z_std_expect"],
                                                                           s["z_std_se"]))
    if s.get("n_var"):
        if abs(s["log_var_ratio"]) > sigma * s["log_var_ratio_se"] + 0.003:
            bad.append("pixel variances differ: mean log ratio %.4f (se %.4f)" % (s["log_var_ratio"],
                                                                                 s["log_var_ratio_se"]))
        if abs(s["fano_a"] - s["fano_b"]) > sigma * s["fano_se"] + 0.003:
            bad.append("core Fano factors %.4f vs %.4f (se %.4f)" % (s["fano_a"], s["fano_b"], s["fano_se"]))
        if require_subpoisson is not None and not (s["fano_a"] < require_subpoisson and s["fano_b"] < require_subpoisson):
            bad.append("core Fano factors %.3f / %.3f are not sub-Poisson (< %.2f)" % (s["fano_a"], s["fano_b"],
                                                                                     require_subpoisson))
    if abs(s["total_a"] - s["total_b"]) > sigma * s["total_se"] + 0.5:
        bad.append("electrons kept on the frame: %.1f vs %.1f (se %.2f)" % (s["total_a"], s["total_b"], s["total_se"]))
    if s["row0"] != 0 or s["col0"] != 0:
        bad.append("row 0 / column 0 populated: %g / %g" % (s["row0"], s["col0"]))
    if "wing_chi2" in s and s["wing_dof"] > 0:
        dof = s["wing_dof"]
        if s["wing_chi2"] > dof + sigma * np.sqrt(2.0 * dof):
            bad.append("wings beyond 12 rows: chi2 %.1f for %d cells" % (s["wing_chi2"], dof))
    return bad


def analytic_moments(counts, x, y, ratio, sl, sh, n):
    """Exact per-pixel mean and variance of the reference thrower's frame for FIXED counts
    (wayne/pyparallel_menu.c:87-108): bin b throws N_b = (int)(counts_b * ratio_b) electrons with sigma_h and
    counts_b - N_b with sigma_l; an electron lands in pixel (r, c), r, c >= 1, with probability
    [Phi((c+1-x)/s) - Phi((c-x)/s)] * [Phi((r+1-y)/s) - Phi((r-y)/s)] (Box-Muller gives independent normals; the C
    cast truncates toward zero and `0 < pos < n` drops row / column 0 and everything off the frame, :91-93).  A
    pixel's count is a sum of independent binomials:
        mean = sum_b N_b P_h + (n_b - N_b) P_l,      var = mean - sum_b N_b P_h^2 + (n_b - N_b) P_l^2.
    Also returned: the variance a thrower would have that drew the sigma of EVERY electron independently with
    probability ratio_b (or per-pixel Poisson counts): var + sum_b n_b r_b (1 - r_b) (P_h - P_l)^2 -- the law the
    device must NOT follow.  -> (mean, var, var_random_split, expected electrons kept on the frame)"""
    from scipy.special import ndtr
    counts = np.asarray(counts, dtype=np.float64)
    x, y, ratio, sl, sh = (np.asarray(a, dtype=np.float64) for a in (x, y, ratio, sl, sh))
    n_wide = np.trunc(counts * ratio)                     # (int)(counts * ratio), counts * ratio >= 0
    n_narrow = counts - n_wide
    edges = np.arange(1, n + 1, dtype=np.float64)         # pixel j in 1..n-1 collects [j, j+1)

    def axis_probs(pos, sig):
        cdf = ndtr((edges[None, :] - pos[:, None]) / sig[:, None])      # (W, n)
        p = np.zeros((pos.size, n))
        p[:, 1:] = np.diff(cdf, axis=1)
        return p

    mean = np.zeros((n, n))
    second = np.zeros((n, n))
    cross = np.zeros((n, n))
    pxh, pyh = axis_probs(x, sh), axis_probs(y, sh)
    pxl, pyl = axis_probs(x, sl), axis_probs(y, sl)
    mean += (pyh * n_wide[:, None]).T @ pxh + (pyl * n_narrow[:, None]).T @ pxl
    second += ((pyh ** 2) * n_wide[:, None]).T @ (pxh ** 2) + ((pyl ** 2) * n_narrow[:, None]).T @ (pxl ** 2)
    # sum_b n_b r (1 - r) (P_h - P_l)^2 with the effective r = N_b / n_b
    r = np.divide(n_wide, counts, out=np.zeros_like(counts), where=counts > 0)
    w = counts * r * (1 - r)
    cross += ((pyh ** 2) * w[:, None]).T @ (pxh ** 2) + ((pyl ** 2) * w[:, None]).T @ (pxl ** 2) \
        - 2.0 * ((pyh * pyl) * w[:, None]).T @ (pxh * pxl)
    var = mean - second
    return mean, var, var + cross, float(mean.sum())


def compare_with_moments(A, mean, var, var_other=None, min_sum=400, min_mean_var=8.0):
    """One-sample figures of an ensemble A[m] against exact moments: z of the pixel means, the mean ratio of sample
    variance to exact variance (pixels with at least `min_mean_var` electrons per frame, where the count is
    near-gaussian and the ratio's standard error is sqrt(2/(M-1))), the total kept on the frame."""
    A = np.asarray(A, dtype=np.float64)
    M = A.shape[0]
    m, v = A.mean(axis=0), A.var(axis=0, ddof=1)
    sel = (mean * M >= min_sum) & (var > 0)
    z = (m[sel] - mean[sel]) / np.sqrt(var[sel] / M)
    out = {"M": M, "n_bright": int(sel.sum()), "z_mean": float(z.mean()), "z_mean_se": 1.0 / np.sqrt(sel.sum()),
           "z_std": float(z.std(ddof=1)), "z_std_se": 1.0 / np.sqrt(2.0 * sel.sum()), "z_max": float(np.abs(z).max())}
    sv = (mean >= min_mean_var) & (var > 0)
    out["n_var"] = int(sv.sum())
    if sv.any():
        # a count of mean lambda has excess kurtosis ~ 1/lambda: var(s^2)/sigma^4 = 2/(M-1) + kurt/M
        kurt = 1.0 / mean[sv]
        se_px = np.sqrt(2.0 / (M - 1) + kurt / M)
        out["var_ratio"] = float((v[sv] / var[sv]).mean())
        out["var_ratio_se"] = float(np.sqrt((se_px ** 2).sum()) / sv.sum())
        if var_other is not None:
            # where the two variance laws are at least 2 % apart (the core of the trace): the ensemble against each
            sc = sv & (var_other >= 1.02 * var)
            out["n_split"] = int(sc.sum())
            if sc.any():
                se_c = float(np.sqrt((se_px[sc[sv]] ** 2).sum()) / sc.sum())
                out["split_ratio_exact"] = float((v[sc] / var[sc]).mean())
                out["split_ratio_other"] = float((v[sc] / var_other[sc]).mean())
                out["split_se"] = se_c
                out["other_over_exact"] = float((var_other[sc] / var[sc]).mean())
    t = A.reshape(M, -1).sum(axis=1)
    out["total"], out["total_expect"] = float(t.mean()), float(mean.sum())
    out["total_se"] = float(np.sqrt(max(t.var(ddof=1), 0.0) / M))
    return out


def wings_against_moments(A, mean, var, x, y, wing=(13, 30)):
    """The wing cells of compare() -- (row offset from the trace, third of the columns) -- of ONE ensemble against the
    exact law: chi-square of the summed counts with the exact variances (covariances between the pixels of a cell are
    negative and tiny out there: the figure is conservative)."""
    A = np.asarray(A, dtype=np.float64)
    M, n = A.shape[0], A.shape[1]
    S = A.sum(axis=0)
    off = trace_row_offsets(x, y, n)
    lo, hi = wing
    thirds = np.array_split(np.arange(n), 3)
    obs, exp, vv = [], [], []
    for d in list(range(-hi, -lo + 1)) + list(range(lo, hi + 1)):
        sel = off == d
        for cols in thirds:
            m = np.zeros_like(sel)
            m[:, cols] = sel[:, cols]
            obs.append(S[m].sum()); exp.append(M * mean[m].sum()); vv.append(M * var[m].sum())
    obs, exp, vv = np.array(obs), np.array(exp), np.array(vv)
    ok = exp >= 50
    z = (obs[ok] - exp[ok]) / np.sqrt(vv[ok])
    return {"wing_exact_chi2": float((z * z).sum()), "wing_exact_dof": int(ok.sum()), "wing_exact_z_mean": float(z.mean()),
            "wing_exact_rel": float((obs[ok].sum() - exp[ok].sum()) / exp[ok].sum()),
            "wing_exact_rel_se": float(np.sqrt(vv[ok].sum()) / exp[ok].sum())}


def check_wings_against_moments(w, sigma=5.0):
    bad = []
    dof = w["wing_exact_dof"]
    if dof > 0 and w["wing_exact_chi2"] > dof + sigma * np.sqrt(2.0 * dof):
        bad.append("wings against the exact law: chi2 %.1f for %d cells" % (w["wing_exact_chi2"], dof))
    if dof > 0 and abs(w["wing_exact_rel"]) > sigma * w["wing_exact_rel_se"] + 1e-4:
        bad.append("electrons in the wings against the exact law: %+.2e (se %.1e)" % (w["wing_exact_rel"],
                                                                                      w["wing_exact_rel_se"]))
    return bad


def check_moments(s, sigma=5.0):
    bad = []
    if abs(s["z_mean"]) > sigma * s["z_mean_se"]:
        bad.append("pixel means off the exact law: mean z = %.4f (se %.4f)" % (s["z_mean"], s["z_mean_se"]))
    if abs(s["z_std"] - 1.0) > sigma * s["z_std_se"] + 0.01:
        bad.append("spread of pixel z against the exact law: %.4f (se %.4f)" % (s["z_std"], s["z_std_se"]))
    if "var_ratio" in s and abs(s["var_ratio"] - 1.0) > sigma * s["var_ratio_se"] + 0.002:
        bad.append("pixel variance / exact variance = %.4f (se %.4f)" % (s["var_ratio"], s["var_ratio_se"]))
    if s.get("n_split"):
        if abs(s["split_ratio_exact"] - 1.0) > sigma * s["split_se"] + 0.002:
            bad.append("core variance / deterministic-split variance = %.4f (se %.4f)" % (s["split_ratio_exact"],
                                                                                          s["split_se"]))
    if abs(s["total"] - s["total_expect"]) > sigma * s["total_se"] + 0.5:
        bad.append("electrons kept on the frame: %.1f, expected %.1f (se %.2f)" % (s["total"], s["total_expect"],
                                                                                  s["total_se"]))
    return bad
