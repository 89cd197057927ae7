"""CPU: the chain that carries the production thrower's law to the reference's, link by link, in DISTRIBUTION
(SURVEY.md section 7 step 4).  tests/test_ensemble_gpu.py puts the device itself beside the reference; here, without a
GPU:

  reference C thrower (oracle/_ref, wayne/pyparallel_menu.c:87-108)
      ~ exact per-pixel moments of its algorithm (tests/ensemble_stats.analytic_moments)      <- pins the closed form
      ~ oracle/split_oracle.c, the CPU statement of the device's default mode                  <- pins the split law
  (device == split_oracle.c on the same counters: tests/test_split_gpu.py)

Ensembles are smaller than on the GPU box (the CPU suite has minutes, not seconds per frame to spend).
"""
import numpy as np
import pytest

import ensemble_stats as es
from conftest import load_golden_psf
from oracle import clib

needs_ref = pytest.mark.skipif(not clib.have_ref(), reason="oracle/_ref not built")


def _inputs(name):
    if name == "bright":
        k = load_golden_psf("s256_t4")
        return k, (k["counts"].astype(np.int64) * 20).astype(np.int32), 40, 120
    k = load_golden_psf("edge_low")
    return k, (k["counts"].astype(np.int64) * 20).astype(np.int32), 64, 200


@needs_ref
@pytest.mark.parametrize("name", ["bright", "edge"])
def test_reference_and_split_oracle_ensembles_follow_the_exact_moments(name):
    k, counts, m_ref, m_split = _inputs(name)
    n = k["nr"]
    args = (counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"])
    mean, var, var_other, _ = es.analytic_moments(*args, n)
    tests = np.random.RandomState(5).randint(0, 100000, m_ref)
    A = np.stack([clib.psf_reference(*args, n, n, int(tests[m]), 1 if m % 2 == 0 else 4).reshape(n, n)
                  for m in range(m_ref)])
    B = np.stack([clib.psf_split_oracle(*args, n, 1963, m, m % 7).reshape(n, n) for m in range(m_split)])
    bad = ["reference: " + b for b in es.check_moments(es.compare_with_moments(A, mean, var, var_other))]
    bad += ["reference: " + b for b in es.check_wings_against_moments(es.wings_against_moments(A, mean, var, k["x"], k["y"]))]
    one = es.compare_with_moments(B, mean, var, var_other)
    bad += ["split oracle: " + b for b in es.check_moments(one)]
    bad += ["split oracle: " + b for b in es.check_wings_against_moments(es.wings_against_moments(B, mean, var, k["x"], k["y"]))]
    bad += ["split oracle vs reference: " + b for b in es.check(es.compare(B, A, k["x"], k["y"]))]
    assert not bad, "; ".join(bad)


def test_the_statistics_see_a_wrong_law():
    # the figures must have teeth: ensembles of the restated thrower with (a) sigma_h 1 % off, (b) the trace 0.004 px off,
    # (c) the sigma of every electron drawn independently instead of N = (int)(counts * ratio) -- each must fail
    k = load_golden_psf("s256_t4")
    n = k["nr"]
    counts = (k["counts"].astype(np.int64) * 10).astype(np.int32)
    base = (k["x"], k["y"], k["ratio"], k["sl"], k["sh"])
    mean, var, var_other, _ = es.analytic_moments(counts, *base, n)
    rs = np.random.RandomState(3)
    M = 40

    def ens(c, x, y, ratio, sl, sh, m0):
        return np.stack([clib.psf_oracle(c, x, y, ratio, sl, sh, n, n, 1000 + 37 * (m0 + m), 1).reshape(n, n)
                         for m in range(M)])

    good = ens(counts, *base, 0)
    assert not es.check_moments(es.compare_with_moments(good, mean, var, var_other))
    assert not es.check_wings_against_moments(es.wings_against_moments(good, mean, var, k["x"], k["y"]))
    wide = ens(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"] * 1.01, 100)
    assert es.check(es.compare(wide, good, k["x"], k["y"]))
    assert es.check_wings_against_moments(es.wings_against_moments(wide, mean, var, k["x"], k["y"]))     # 1 % wider: more in the wings
    assert es.check_moments(es.compare_with_moments(wide, mean, var, var_other))
    moved = ens(counts, k["x"] + 0.004, k["y"] + 0.004, k["ratio"], k["sl"], k["sh"], 200)
    assert es.check_moments(es.compare_with_moments(moved, mean, var, var_other))
    # (c): binomial sigma split, thrown as two calls (all-wide + all-narrow)
    one, zero = np.ones_like(k["ratio"]), np.zeros_like(k["ratio"])
    fr = []
    for m in range(160):
        nw = rs.binomial(counts, np.clip(k["ratio"], 0, 1)).astype(np.int32)
        fr.append(clib.psf_oracle(nw, k["x"], k["y"], one, k["sl"], k["sh"], n, n, 5000 + 37 * m, 1).reshape(n, n) +
                  clib.psf_oracle(counts - nw, k["x"], k["y"], zero, k["sl"], k["sh"], n, n, 90000 - 41 * m, 1
                                  ).reshape(n, n))
    s = es.compare_with_moments(np.stack(fr), mean, var, var_other)
    assert s["n_split"] >= 50
    assert (s["split_ratio_exact"] - 1.0) > 4.0 * s["split_se"], s      # excess variance of the random split shows
    assert abs(s["split_ratio_other"] - 1.0) < 4.0 * s["split_se"] + 0.002, s
