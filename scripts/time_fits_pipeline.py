#!/usr/bin/env python3
"""Exposures/s of a visit written to FITS files (float64 SCI extensions, the reference's layout, exposure.py:133-214),
and where the time of one file goes: device -> pinned host memory (PCIe), float32 -> big-endian float64 (format),
the write itself (disk).  GPU only.

    python scripts/time_fits_pipeline.py [config=cfg4] [exposures=24]
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from wayne_amd import fitsio, visit as wv  # noqa: E402
from wayne_amd.exposure import Exposure  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
v = helpers.make_visit(name, n_exposures=n + 2)
out = tempfile.mkdtemp(prefix="wayne_fits_")
try:
    r = wv.VisitRunner(v, 0, out_dir=out)
    r.run([0, 1])
    t = time.perf_counter()
    r.run(range(2, n + 2))
    dt = time.perf_counter() - t
    size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    per_file = size / (n + 2)
    print("%s: %d exposures -> FITS in %.2f s = %.1f exposures/s (%.0f MB/s, %d files of %.1f MB; %d writer threads)" % (
        name, n, dt, n / dt, per_file * n / dt / 1e6, len(os.listdir(out)), per_file / 1e6,
        int(os.environ.get("WAYNE_FITS_THREADS", "0")) or min(16, os.cpu_count() or 4)))
    # the parts, one at a time
    r2 = wv.VisitRunner(v, 0)
    t = time.perf_counter()
    got = r2.run(range(2, n + 2), keep=False)
    d_host = (time.perf_counter() - t) / n
    reads = r2.run([0], keep=True)[0]
    gen = r2.generator(0)
    gen.build_descriptor(None, **v.frame_kwargs(0))
    exp = Exposure(gen.detector, gen.grism, None, gen.exp_info)
    for i in range(reads.shape[0]):
        exp.add_read(reads[i], {"cumulative_exp_time": 1.0 * i, "read_exp_time": 1.0, "CRPIX1": 0})
    t_fmt, t_wr = [], []
    for rep in range(5):
        t0 = time.perf_counter()
        cube = np.empty(reads.shape, dtype=">f8")
        for i in range(reads.shape[0]):
            cube[i] = reads[i]
        t1 = time.perf_counter()
        fitsio.write_pieces(os.path.join(out, "probe_%d.fits" % rep), [memoryview(cube.reshape(-1)).cast("B")])
        t2 = time.perf_counter()
        t_fmt.append(t1 - t0)
        t_wr.append(t2 - t1)
    t0 = time.perf_counter()
    for rep in range(3):
        exp.generate_fits(out, "whole_%d.fits" % rep)
    d_file = (time.perf_counter() - t0) / 3
    print("parts per exposure: device -> pinned host (kernels + PCIe, pipelined) %.1f ms = %.0f/s; float32 -> big-endian "
          "float64 (one thread) %.1f ms; write of %.0f MB (one thread) %.1f ms = %.2f GB/s; generate_fits whole, one "
          "thread %.1f ms" % (d_host * 1e3, 1.0 / d_host, np.median(t_fmt) * 1e3, cube.nbytes / 1e6, np.median(t_wr) * 1e3,
                              cube.nbytes / np.median(t_wr) / 1e9, d_file * 1e3))
finally:
    shutil.rmtree(out, ignore_errors=True)
