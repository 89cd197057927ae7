#!/usr/bin/env python3
"""k_ramp's time against the number of reads it works through (GPU only): t(R) = t0 + R t1 separates what a launch pays
once per pixel (launch, alias tables -> LDS, stream seeding, once-per-pixel planes, zero read) from what it pays per
read.  Same exposure (cfg4), same kernel instantiation, R = 1 ... 15 through the measurement knob WAYNE_RAMP_READS --
which only a TIMING BUILD of the library looks at (-DWAYNE_TIMING_KNOBS: the reads left out keep stale planes and
uncleared accumulators, so the shipped library ignores the variable).  This script builds that library
(ab/timing_knobs.so, hipcc, ~25 s) and loads it through WAYNE_HIP_LIB.

    python scripts/ramp_vs_reads.py [launches per point = 30]

A straight line is the healthy picture.  Round 4 found t(3) = 20 us, t(4) = 49 us, flat to t(12): ONE wave per launch
walking a 512-step search (a sky draw whose uniform fell into the rounding residue of its float32 cdf) while every
other wave had finished -- see HISTORY.md section 9.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import build as _wb  # noqa: E402
os.environ["WAYNE_ALLOW_FLAGGED_LIB"] = "1"      # a timing build, on purpose (wayne_amd/_lib.py refuses one otherwise)
os.environ["WAYNE_HIP_LIB"] = _wb.build_variant(["-DWAYNE_TIMING_KNOBS"] + os.environ.get("WAYNE_CXXFLAGS", "").split(),
                                                os.path.join(ROOT, "ab", "timing_knobs.so"))
from wayne_amd import calibration, detector, engine, grism, synthetic  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cal = calibration.CalibrationSet.synthetic(11)
det = detector.WFC3_IR()
gr = grism.G141(cal)
v = synthetic.Visit("cfg4", det, gr, cal, n_exposures=1)
eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
ctx = eng.ctx
eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed)
ctx.upload(0, eg.build_descriptor(eng, out_dtype=np.float32, **v.frame_kwargs(0)))
ctx.run(0)
ctx.synchronize()
rows = []
for R in [15] + list(range(1, 16)):
    ctx.set_knob("ramp_reads", R)
    ctx.run(0)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(reps):
        ctx.run(0)
    p = ctx.profile_get()
    ctx.profile_enable(False)
    t = p["k_ramp"]["ms"] / p["k_ramp"]["launches"] * 1e3
    rows.append((R, t))
    print("R = %2d   k_ramp %.2f us" % (R, t), flush=True)
ctx.set_knob("ramp_reads", None)
R = np.array([r for r, _ in rows[1:]], dtype=float)
T = np.array([t for _, t in rows[1:]])
t1, t0 = np.polyfit(R, T, 1)
print("fit: t0 = %.2f us per launch, t1 = %.2f us per read; largest residual %.2f us" % (
    t0, t1, float(np.abs(T - (t0 + t1 * R)).max())))
