#!/usr/bin/env python3
"""The inner drop-in boundary by itself: wayne.pyparallel.apply_psf (pyparallel.pyx:14-38) for ONE sub-sample of the
benchmarked workload -- host arrays in, the (NR, NC) electron frame out, PCIe and launch included -- in replay mode
(the reference's frame bit for bit), per-electron Philox mode and the split thrower, through the C ABI
(wayne_psf_apply) and through the Python shim (which adds the reference's float64 conversion of the frame).

    python scripts/bench_psf_apply.py [calls=30]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import _lib, calibration, detector, grism, pyparallel, synthetic  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cal = calibration.CalibrationSet.synthetic(11)
gr = grism.G141(cal)
v = synthetic.Visit("cfg4", detector.WFC3_IR(), gr, cal, n_exposures=1)
W = v.wl.size
rng = np.random.default_rng(3)
# one sub-sample of cfg4: 1e9 / 128 electrons over the bins with the spectrum's shape, trace across the frame
shape = v.stellar_flux / v.stellar_flux.sum()
counts = rng.poisson(shape * v.E / v.K).astype(np.int32)
x = np.linspace(560.0, 700.0, W)
y = 300.0 + 0.01 * (x - 560.0)
ratio = np.polyval(gr.psf_ratio_poly, v.wl) if not callable(gr.psf_ratio_poly) else gr.psf_ratio_poly(v.wl)
sl = np.polyval(gr.psf_sigmal_poly, v.wl) if not callable(gr.psf_sigmal_poly) else gr.psf_sigmal_poly(v.wl)
sh = np.polyval(gr.psf_sigmah_poly, v.wl) if not callable(gr.psf_sigmah_poly) else gr.psf_sigmah_poly(v.wl)
ctx = _lib.default_context(0)
E = int(counts.sum())
print("one sub-sample: %d bins, %d electrons, frame 1014 x 1014" % (W, E))
for name, mode in (("replay (bit-exact)", _lib.RNG_REPLAY), ("per-electron Philox", _lib.RNG_PHILOX), ("split", _lib.RNG_SPLIT)):
    for j in range(3):
        ctx.psf_apply(counts, x, y, ratio, sl, sh, 1014, 1014, 7 + j, 4, mode)
    t0 = time.perf_counter()
    for j in range(n_calls):
        f = ctx.psf_apply(counts, x, y, ratio, sl, sh, 1014, 1014, 7 + j, 4, mode)
    dt = (time.perf_counter() - t0) / n_calls
    assert int(f.sum()) > 0.97 * E
    t0 = time.perf_counter()
    for j in range(max(n_calls // 3, 3)):
        pyparallel.apply_psf(counts, x, y, ratio, sl, sh, 1014, 1014, 7 + j, 4, rng_mode=mode)
    dt_py = (time.perf_counter() - t0) / max(n_calls // 3, 3)
    print("%-22s C ABI %.3f ms per call = %.3g electrons/s; Python shim (float64 frame) %.3f ms" % (name, dt * 1e3, E / dt, dt_py * 1e3))
