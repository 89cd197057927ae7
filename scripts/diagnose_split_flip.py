#!/usr/bin/env python3
"""Which draw moved the electrons?  (GPU only; test infrastructure: uses oracle/.)

Regenerates case `index` of scripts/soak_split.py (same rng, same order), runs the device's split thrower in
production math (as the soak does) and with exact samplers, and oracle/split_oracle.c with every binomial call
recorded.  Lane-thrown electrons cancel in (device - oracle), so per bin the difference of the two frames over the
bin's 13 x 13 window IS the difference of the two multinomial draws; the script

  1. lists the bins whose windows differ and the electrons moved in each,
  2. rebuilds the device's column counts of such a bin (oracle's + the column sums of the difference) and names
     the FIRST step of the chain whose count differs: (n, p, oracle's k, device's k),
  3. re-runs the oracle with that one call's p changed by +-1, 2, 4, ... ulp and reports the smallest change that
     reproduces the device's count at that step -- a last-bit flip shows up at a few ulp; a biased probability
     (a wrong upper_tail / div_ path) would need a change orders of magnitude above the arithmetic's error.

    python scripts/diagnose_split_flip.py [index=217] [seed=1]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from oracle import clib  # noqa: E402
from wayne_amd import _lib  # noqa: E402
import soak_split  # noqa: E402

R = 6   # window half-width (kNarrowR)


def main():
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 217
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for i in range(index + 1):
        counts, x, y, ratio, sl, sh, N = soak_split.case(rng)
        seed, exp, sub = int(rng.integers(0, 2**31)), int(rng.integers(0, 100)), int(rng.integers(0, 3000))
    W = counts.size
    print("case %d: W=%d N=%d electrons=%d ratio=%.2f seed=%d exposure=%d subsample=%d" % (
        index, W, N, counts.sum(), ratio[0], seed, exp, sub))
    ctx = _lib.Context(0)
    want, calls = clib.psf_split_trace(counts, x, y, ratio, sl, sh, N, seed, exp, sub)
    want = want.reshape(N, N).astype(np.int64)
    frames = {}
    for tag, exact in (("production math", False), ("exact samplers", True)):
        got = ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, seed, rng_mode=_lib.RNG_SPLIT, exposure=exp,
                            subsample=sub, exact_samplers=exact).reshape(N, N).astype(np.int64)
        frames[tag] = got
        print("%-16s moved %d of %d electrons (%.2e)" % (tag, np.abs(got - want).sum() // 2, want.sum(),
                                                        np.abs(got - want).sum() / 2 / max(want.sum(), 1)))
    diff = frames["production math"] - want
    if not diff.any():
        print("no difference to diagnose")
        return
    # 1. bins whose windows hold a difference (bins of this case are far apart or we say so)
    ic, jc = np.floor(x).astype(int), np.floor(y).astype(int)
    nw = np.minimum(np.maximum((counts * ratio).astype(np.int64), 0), counts)
    narrow = counts - nw
    split = (narrow >= 32) & (sl > 0.05) & (sl * 6.5 <= R)
    ys, xs = np.nonzero(diff)
    print("differing pixels: %d, x %d..%d, y %d..%d" % (ys.size, xs.min(), xs.max(), ys.min(), ys.max()))
    hit = [b for b in range(W) if split[b] and np.abs(diff[max(jc[b] - R, 0):jc[b] + R + 1, max(ic[b] - R, 0):ic[b] + R + 1]).sum()]
    print("split bins whose 13 x 13 window holds a difference: %d of %d %s" % (len(hit), int(split.sum()), hit[:12]))
    # A bin of a group that does not pool its rows draws its chain from its own stream and its own numbers alone, so
    # the bin can be re-run BY ITSELF (every other count zero: same bin index, same stream) on both sides, and the
    # difference of the two frames is then this bin's alone even where windows overlap in the full case.
    hit = hit[:6]
    counts_all = counts
    for b in hit:
        counts = np.zeros_like(counts_all)
        counts[b] = counts_all[b]
        want_b, calls = clib.psf_split_trace(counts, x, y, ratio, sl, sh, N, seed, exp, sub)
        got_b = ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, seed, rng_mode=_lib.RNG_SPLIT, exposure=exp,
                              subsample=sub).reshape(N, N).astype(np.int64)
        diff_b = got_b - want_b.reshape(N, N)
        win = diff_b[jc[b] - R:jc[b] + R + 1, ic[b] - R:ic[b] + R + 1]
        print("\nbin %d alone: x=%.4f y=%.4f sigma_l=%.4f narrow=%d wide=%d, moved %d (in its window: |diff| = %d, net %d)" % (
            b, x[b], y[b], sl[b], narrow[b], nw[b], np.abs(diff_b).sum() // 2, np.abs(win).sum(), win.sum()))
        if not win.any():
            print("  alone, the bin agrees: in the full case its group pools its rows (the flip is in a pooled chain)")
            continue
        # 2. the bin's chain in the oracle's trace, in drawing order: column c (centre-out), then that column's rows
        # (centre-out) until the column is used up; the device's count of a call = the oracle's + the difference frame
        off = lambda c: 0 if c == 0 else ((c + 1) // 2 if c % 2 else -(c // 2))
        i, left, first = 0, float(narrow[b]), None
        col_diff = win.sum(axis=0)
        c = 0
        while left > 0 and c < 2 * R + 1 and i < len(calls):
            n_c, p_c, k_c = (float(v) for v in calls[i])
            assert n_c == left, (i, n_c, left)
            dk = int(col_diff[R + off(c)])
            if dk != 0 and first is None:
                first = (i, "column step %d (x = ic%+d)" % (c, off(c)), n_c, p_c, k_c, dk)
            i += 1
            left -= k_c
            m, r = k_c, 0
            while m > 0 and r < 2 * R + 1:
                n_r, p_r, k_r = (float(v) for v in calls[i])
                dk = int(win[R + off(r), R + off(c)])
                if dk != 0 and first is None:
                    first = (i, "row step %d (y = jc%+d) of column step %d (x = ic%+d)" % (r, off(r), c, off(c)), n_r, p_r, k_r, dk)
                m -= k_r
                i += 1
                r += 1
            c += 1
        if first is None:
            print("  (no differing call found in the bin's chain)")
            continue
        ci, where, n_c, p_c, k_c, dk = first
        print("  first call of the chain whose count differs: call %d, %s" % (ci, where))
        # 3. smallest relative change of that call's p that gives the device's count
        ulp = float(np.spacing(np.float32(p_c)) / p_c)
        found = None
        for mult in (1, 2, 4, 8, 16, 32, 64, 256, 1024, 16384):
            for sign in (1, -1):
                _, calls2 = clib.psf_split_trace(counts, x, y, ratio, sl, sh, N, seed, exp, sub, perturb_at=ci,
                                                 perturb_rel=sign * mult * ulp)
                if calls2[ci, 2] == k_c + dk:
                    found = (sign * mult, calls2[ci, 1])
                    break
            if found:
                break
        sd = np.sqrt(n_c * p_c * (1 - p_c))
        print("  the draw: Binomial(n=%.0f, p=%.9g) -> oracle %.0f, device %.0f (a difference of %.2f sigma of the draw)" % (
            n_c, p_c, k_c, k_c + dk, abs(dk) / max(sd, 1e-30)))
        if found:
            print("  changing p by %+d ulp (p = %.9g, relative %.1e) makes the oracle draw the device's count: "
                  "a last-bit flip" % (found[0], found[1], abs(found[0]) * ulp))
        else:
            print("  no change of p up to 16384 ulp reproduces the device's count: NOT a last-bit flip")


if __name__ == "__main__":
    main()
