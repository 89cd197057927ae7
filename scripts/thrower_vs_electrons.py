#!/usr/bin/env python3
"""The thrower kernels' time against the number of electrons of the exposure (GPU only): t(E) = t0 + E t1 separates
what a launch pays whatever it throws (launch, prologues, tile clears and flushes, the tail of the grid) from what an
electron costs.  cfg4's geometry with the stellar flux scaled.

    python scripts/thrower_vs_electrons.py [launches per point = 20]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import calibration, detector, engine, grism, synthetic  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cal = calibration.CalibrationSet.synthetic(11)
det = detector.WFC3_IR()
gr = grism.G141(cal)
rows = []
for E in (0.0625e9, 0.125e9, 0.25e9, 0.5e9, 1e9, 1.5e9, 2e9):
    v = synthetic.Visit("cfg4", det, gr, cal, n_exposures=1, E=E)
    eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed)
    ctx.upload(0, eg.build_descriptor(eng, out_dtype=np.float32, **v.frame_kwargs(0)))
    ctx.run(0)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(reps):
        ctx.run(0)
    p = ctx.profile_get()
    ctx.profile_enable(False)
    t = {k: p[k]["ms"] / max(p[k]["launches"], 1) * 1e3 for k in ("k_prep_sub", "k_lane", "k_narrow", "k_ramp")}
    rows.append((p["electrons"] / reps, t))
    print("E = %.3g   " % (p["electrons"] / reps) + "  ".join("%s %.1f us" % (k, x) for k, x in t.items()), flush=True)
E = np.array([r[0] for r in rows])
for k in ("k_lane", "k_narrow", "k_prep_sub"):
    T = np.array([r[1][k] for r in rows])
    t1, t0 = np.polyfit(E, T, 1)
    print("%s: t0 = %.1f us per launch, %.1f us per 1e9 electrons" % (k, t0, t1 * 1e9))
