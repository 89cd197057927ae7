#!/usr/bin/env python3
"""The narrow PSF component as multinomials (k_narrow), cell by cell at 16 x the suite's statistics: 3.2e9 all-narrow
electrons from one position, every column and row within 8 px of it against the gaussian's exact pixel masses
(pyparallel_menu.c:87-108), for bins that pool their row chains and bins that do not.  The suite's
test_narrow_electrons_fill_their_window_with_the_gaussians_masses runs one sixteenth of this; profiles/r05/narrow_window_probe.txt.

    python scripts/narrow_probe.py        # on the GPU box, ~1 s
"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.stats import norm
from wayne_amd import _lib
ctx = _lib.Context(0)
B, n_each, cx, cy, N = 50000, 4000, 507.3, 507.6, 1014
one = np.ones(B); counts = np.full(B, n_each, dtype=np.int32)
edges = np.arange(N + 1, dtype=float)
for sigmas in ((0.7,), (0.7, 0.85)):
    sl = np.resize(np.asarray(sigmas), B)
    f = np.zeros((N, N)); calls = 16
    for e in range(calls):
        f += np.asarray(ctx.psf_apply(counts, cx * one, cy * one, 0.0 * one, sl, 5.5 * one, N, N, 1234 + e, 1, rng_mode=_lib.RNG_SPLIT, exposure=e), dtype=float).reshape(N, N)
    total = float(B) * n_each * calls
    for axis, c in (("columns", cx), ("rows", cy)):
        got = f.sum(axis=0 if axis == "columns" else 1)
        p = np.mean([np.diff(norm.cdf((edges - c) / s_)) for s_ in sigmas], axis=0)
        want = total * p
        centre = np.arange(N) + 0.5 - c
        sel = np.abs(centre) < 8
        print(sigmas, axis)
        for i in np.nonzero(sel)[0]:
            if want[i] > 0.5 or got[i] > 0:
                print("  offset %+5.1f got %12.0f want %14.2f z %+6.2f" % (centre[i], got[i], want[i], (got[i] - want[i]) / np.sqrt(max(want[i], 1e-9))))
