#!/usr/bin/env python3
"""Soak of the replay (bit-exact) thrower against the CPU restatement of the reference's PSF(): random bins, frames,
PSF widths from 1e-3 to 200 px, thread counts 1..16, positions on and off the frame and on pixel boundaries.  Every
frame must equal the oracle's bit for bit (and, where oracle/_ref is built, the compiled reference C's).

    python scripts/soak_replay.py [cases=300] [seed=1]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clib  # noqa: E402  (the checker: test infrastructure)
from wayne_amd import _lib  # noqa: E402

def case(rng):
    """One random input of PSF() inside the reference's valid domain."""
    W = int(rng.integers(1, 4000))
    N = int(rng.choice([16, 33, 64, 100, 256, 512, 1014]))
    threads = int(rng.integers(1, 17))
    lam = 10.0 ** rng.uniform(-1, 2.7)
    counts = rng.poisson(lam, W).astype(np.int32)
    if rng.random() < 0.2:
        counts[rng.integers(0, W)] = int(rng.integers(100000, 2000000))
    x = rng.uniform(-10, N + 10, W)
    y = rng.uniform(-10, N + 10, W)
    snap = rng.random(W) < 0.15
    x[snap] = np.round(x[snap]) + rng.choice([0.0, 0.5, 2.0 ** -30, -2.0 ** -30], int(snap.sum()))
    y[snap] = np.round(y[snap])
    scale = 10.0 ** rng.uniform(-3, 0.5)
    sl = scale * rng.uniform(0.3, 1.5, W)
    sh = sl * rng.uniform(1.0, 60.0, W)
    # (psf_ratio in [0, 1], ends included: outside it the reference's two loops throw more electrons than it drew
    # normals for -- pyparallel_menu.c:89-106 reads past its array -- and the restatement clamps instead)
    ratio = np.clip(rng.uniform(-0.1, 1.1, W), 0.0, 1.0)
    test = int(rng.integers(0, 2 ** 31 - 1 - 25234 - 17 * 16))
    return counts, x, y, ratio, sl, sh, N, test, threads


def run(ctx, n_cases, seed, verbose=True):
    """-> (frames that differ, electrons thrown)"""
    rng = np.random.default_rng(seed)
    bad = electrons = 0
    for i in range(n_cases):
        counts, x, y, ratio, sl, sh, N, test, threads = case(rng)
        if int(counts.sum()) * threads >= 2 ** 31:
            continue
        want = clib.psf_oracle(counts, x, y, ratio, sl, sh, N, N, test, threads)
        got = ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, test, threads, rng_mode=0)
        ok = np.array_equal(got, want)
        if ok and clib.have_ref() and i % 5 == 0:
            ok = np.array_equal(got, clib.psf_reference(counts, x, y, ratio, sl, sh, N, N, test, threads))
        electrons += int(counts.sum())
        if not ok:
            bad += 1
            if verbose:
                print("case %d DIFFERS: W=%d N=%d threads=%d electrons=%d" % (i, counts.size, N, threads, counts.sum()), flush=True)
    return bad, electrons


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad, electrons = run(_lib.default_context(0), n_cases, seed)
    print("soak_replay %s: %d cases, %.3g electrons, %d frames differ" % ("ok" if bad == 0 else "FAILED", n_cases, electrons, bad))
    sys.exit(1 if bad else 0)
