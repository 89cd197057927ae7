#!/usr/bin/env python3
"""Exposures/s of a visit written to FITS files (float64 SCI extensions, the reference's layout)."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from wayne_amd import visit as wv  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
v = helpers.make_visit(name, n_exposures=n + 2)
out = tempfile.mkdtemp(prefix="wayne_fits_")
try:
    r = wv.VisitRunner(v, 0, out_dir=out)
    r.run([0, 1])
    t = time.perf_counter()
    r.run(range(2, n + 2))
    dt = time.perf_counter() - t
    size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print("%s: %d exposures -> FITS in %.2f s = %.1f exposures/s (%.0f MB/s, %d files)" % (
        name, n, dt, n / dt, size / (n + 2) * n / dt / 1e6, len(os.listdir(out))))
finally:
    shutil.rmtree(out, ignore_errors=True)
