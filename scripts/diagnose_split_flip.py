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
    # only a bin whose window no other bin's electrons reach can be read off the difference frame
    def lonely(b):
        reach = R + np.ceil(6.8 * np.where(nw > 0, sh, 0.0)).astype(int) + 1     # other bins' wide electrons too
        o = np.arange(W) != b
        return not np.any(o & (counts > 0) & (np.abs(ic - ic[b]) <= R + reach) & (np.abs(jc - jc[b]) <= R + reach))
    hit = [b for b in hit if lonely(b)][:6]
    print("of these, bins that stand alone (analysed below):", hit)
    for b in hit:
        win = diff[jc[b] - R:jc[b] + R + 1, ic[b] - R:ic[b] + R + 1]
        print("\nbin %d: x=%.4f y=%.4f sigma_l=%.4f narrow=%d, |diff| in window = %d (net %d)" % (
            b, x[b], y[b], sl[b], narrow[b], np.abs(win).sum(), win.sum()))
        # 2. the bin's calls in the oracle's trace: groups of 16 bins run in order; inside a non-pooling group each
        # split bin's chain is a run of calls starting with n = narrow[b]
        starts = [i for i in range(len(calls)) if calls[i, 0] == np.float32(narrow[b])]
        cand = None
        for s in starts:
            # a column chain: consecutive column calls have n decreasing by the previous result, with row chains between
            cand = s
            break
        if cand is None:
            print("  (could not locate the bin's chain in the trace)")
            continue
        # walk the chain: column call, then its row calls until the column is used up
        i, left, cols = cand, float(narrow[b]), []
        c = 0
        while left > 0 and c < 2 * R + 1 and i < len(calls):
            n_c, p_c, k_c = calls[i]
            assert n_c == np.float32(left), (i, n_c, left)
            cols.append((i, c, n_c, p_c, k_c))
            left -= float(k_c)
            i += 1
            m = float(k_c)
            r = 0
            while m > 0 and r < 2 * R + 1:
                m -= float(calls[i, 2])
                i += 1
                r += 1
            c += 1
        offs = [0 if c == 0 else ((c + 1) // 2 if c % 2 else -(c // 2)) for _, c, _, _, _ in cols]
        col_diff = win.sum(axis=0)       # per column of the window, x = ic - R + j
        first = None
        for (ci, c, n_c, p_c, k_c), off in zip(cols, offs):
            dk = int(col_diff[R + off])
            mark = ""
            if dk != 0 and first is None:
                first = (ci, c, n_c, p_c, k_c, dk)
                mark = "   <-- first column count that differs"
            print("  column step %2d (x = ic%+d): n=%8.0f p=%.9g  oracle k=%8.0f  device k=%8.0f%s" % (
                c, off, n_c, p_c, k_c, k_c + dk, mark))
        if first is None:
            print("  column counts agree: the difference is inside a row chain")
            continue
        ci, c, n_c, p_c, k_c, dk = first
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
