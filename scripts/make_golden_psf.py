#!/usr/bin/env python3
"""Generate tests/golden/psf_*.npz from the REFERENCE's own compiled C kernel.

Runs in the build container only (needs /root/reference to build
oracle/_ref/libwayne_ref_psf.so via oracle/Makefile).  Each fixture holds the
inputs of one PSF() call (wayne/pyparallel_menu.c:10) and the int32 frame the
reference returned, stored sparsely (flat index, value).  The fixtures are
data; no reference source is stored.

    python scripts/make_golden_psf.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clib  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# G141 PSF polynomials evaluated on a wavelength grid are what the reference
# feeds PSF() (grism.py:85-90); numbers typed from the published coefficients.
P_RATIO = [-0.25063428, 0.8332488, -0.80546074, 0.39896516]
P_SIGL = [0.69245668, -2.1043046, 2.22284446, -0.29689335]
P_SIGH = [2.90366189, -8.81859432, 8.96049229, 2.254503]


def spectrum_case(W, N, mean_count, seed, x0, x1, slope=0.0104, y0=None):
    rng = np.random.RandomState(seed)
    wl = np.linspace(1.0, 1.75, W)
    counts = rng.poisson(mean_count * (0.4 + np.sin(np.linspace(0, np.pi, W)) ** 2), W).astype(np.int32)
    x = np.linspace(x0, x1, W)
    y = (N / 2.0 if y0 is None else y0) + slope * (x - x0)
    return dict(counts=counts, x=x, y=y, ratio=np.polyval(P_RATIO, wl),
                sl=np.polyval(P_SIGL, wl), sh=np.polyval(P_SIGH, wl), nr=N, nc=N)


def cases():
    c = {}
    c["s64_t1"] = dict(spectrum_case(200, 64, 40, 11, 8.3, 55.2), test=7, threads=1)
    c["s64_t3"] = dict(spectrum_case(301, 64, 25, 12, 6.0, 58.0), test=12345, threads=3)
    c["s128_t2"] = dict(spectrum_case(1000, 128, 30, 13, 10.5, 118.2), test=99999, threads=2)
    # the example-visit shape: 4494 bins on a 256 frame, threads=4 (yml:6)
    c["s256_t4"] = dict(spectrum_case(4494, 256, 40, 14, 25.4, 200.9, y0=78.2), test=31337, threads=4)
    c["s256_t8"] = dict(spectrum_case(4494, 256, 12, 15, 25.4, 200.9, y0=150.7), test=5, threads=8)
    # spectrum running off the frame: left/bottom edge, truncation of (-1,1) to 0 is rejected
    e = spectrum_case(400, 64, 60, 16, -6.0, 30.0, y0=1.2)
    c["edge_low"] = dict(e, test=1, threads=2)
    e = spectrum_case(400, 64, 60, 17, 40.0, 70.0, y0=62.5)
    c["edge_high"] = dict(e, test=2, threads=2)
    # zero-count bins interleaved, and a single bin
    z = spectrum_case(150, 64, 30, 18, 10.0, 50.0)
    z["counts"][::3] = 0
    z["counts"][:10] = 0
    c["zeros"] = dict(z, test=3, threads=4)
    one = spectrum_case(2, 32, 0, 19, 15.5, 15.5)
    one["counts"] = np.array([977, 0], dtype=np.int32)
    c["single_bin"] = dict(one, test=4, threads=3)
    # fewer electrons than threads (empty OpenMP partitions)
    few = spectrum_case(5, 32, 0, 20, 10.0, 20.0)
    few["counts"] = np.array([1, 0, 2, 0, 1], dtype=np.int32)
    c["few_electrons"] = dict(few, test=9, threads=8)
    # nothing at all
    none = spectrum_case(5, 32, 0, 21, 10.0, 20.0)
    none["counts"] = np.zeros(5, dtype=np.int32)
    c["no_electrons"] = dict(none, test=9, threads=2)
    # ratio edge cases: all wide, all narrow
    r = spectrum_case(120, 64, 50, 22, 12.0, 52.0)
    r["ratio"] = np.where(np.arange(120) % 2 == 0, 1.0, 0.0)
    c["ratio_01"] = dict(r, test=77, threads=2)
    # full array, 1.8e6 electrons (the BASELINE.md sizing probe shape)
    big = spectrum_case(4494, 1014, 400, 23, 430.2, 610.7, y0=507.3)
    c["s1014_t4"] = dict(big, test=4242, threads=4)
    return c


def main():
    clib.build()
    if not clib.have_ref():
        raise SystemExit("oracle/_ref not built: /root/reference absent")
    os.makedirs(OUT, exist_ok=True)
    for name, k in cases().items():
        frame = clib.psf_reference(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"],
                                   k["nr"], k["nc"], k["test"], k["threads"])
        idx = np.flatnonzero(frame).astype(np.int32)
        val = frame[idx].astype(np.int32)
        path = os.path.join(OUT, "psf_%s.npz" % name)
        np.savez_compressed(path, counts=k["counts"], x=k["x"], y=k["y"], ratio=k["ratio"], sl=k["sl"],
                            sh=k["sh"], nr=k["nr"], nc=k["nc"], test=k["test"], threads=k["threads"],
                            idx=idx, val=val)
        print("%-16s W=%5d N=%4d T=%d electrons=%8d kept=%8d -> %s (%d B)" % (
            name, k["counts"].size, k["nr"], k["threads"], int(k["counts"].sum()), int(val.sum()),
            os.path.relpath(path, ROOT), os.path.getsize(path)))


if __name__ == "__main__":
    main()
