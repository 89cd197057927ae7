#!/usr/bin/env python3
"""Randomised soak of the default (split) thrower against oracle/split_oracle.c on the same counters (GPU only):
random numbers of bins, trace-like and scattered positions, thin and dense bins, sigma ranges that make groups
pool their rows, fall back, or mix inside a wave.  Test infrastructure (uses oracle/).

    python scripts/soak_split.py [cases=40] [seed=1]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clib  # noqa: E402
from wayne_amd import _lib  # noqa: E402


def case(rng):
    W = int(rng.choice([17, 100, 513, 1500, 4494, 6000]))
    N = int(rng.choice([128, 256, 512]))
    kind = rng.integers(0, 4)
    x = np.sort(rng.uniform(5, N - 5, W)) if kind == 0 else 8.3 + (N - 20) * np.arange(W) / W
    slope = rng.choice([0.0, 0.003, 0.012, 0.2])
    y = N / 2 + rng.uniform(-0.5, 0.5) + slope * (x - x[0]) + (rng.normal(0, 0.3, W) if kind == 1 else 0)
    y = np.clip(y, 2, N - 2)
    dense = rng.integers(0, 3)
    counts = rng.integers(0, [40, 4000, 40000][dense], W).astype(np.int32)
    if kind == 2:
        counts[rng.integers(0, W, W // 5)] = rng.integers(0, 25, W // 5)
    ratio = np.full(W, rng.choice([0.0, 0.1, 0.27, 0.9]))
    sl = np.linspace(*sorted(rng.uniform(0.3, 1.1, 2)), W) if kind != 3 else rng.uniform(0.45, 0.95, W)
    sh = np.linspace(*sorted(rng.uniform(1.5, 6.0, 2)), W)
    return counts, x, y, ratio, sl, sh, N


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = _lib.Context(0)
    worst = 0.0
    for i in range(n_cases):
        counts, x, y, ratio, sl, sh, N = case(rng)
        seed, exp, sub = int(rng.integers(0, 2**31)), int(rng.integers(0, 100)), int(rng.integers(0, 3000))
        want = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, seed, exp, sub)
        got = ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, seed, rng_mode=_lib.RNG_SPLIT, exposure=exp, subsample=sub)
        again = ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, seed, rng_mode=_lib.RNG_SPLIT, exposure=exp, subsample=sub)
        total = int(want.sum())
        moved = int(np.abs(got.astype(np.int64) - want).sum()) // 2
        frac = moved / max(total, 1)
        worst = max(worst, frac)
        # (tests/helpers.py split_moved_bound: a rate term + room for ONE flipped chain; HISTORY.md section 6)
        ok = (np.array_equal(got, again) and abs(int(got.sum()) - total) <= 2 + total // 100000 and
              moved <= 2 + 1e-4 * total + 3 * np.sqrt(16.0 * counts.max()))
        print("case %2d W=%4d N=%3d electrons=%9d moved=%6d (%.1e) %s" % (i, counts.size, N, total, moved, frac,
                                                                          "ok" if ok else "FAIL"), flush=True)
        if not ok:
            raise SystemExit(1)
    print("soak_split ok: %d cases, worst moved fraction %.1e" % (n_cases, worst))


if __name__ == "__main__":
    main()
