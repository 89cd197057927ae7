#!/usr/bin/env python3
"""k_lane / k_narrow / k_prep_sub against the number of sub-samples K at a fixed number of electrons PER SUB-SAMPLE (GPU
only): cfg4 launches 9 chunks x K workgroups of 512 bins; four `k_lane` workgroups fit a CU (36 KB tiles), 1024 on the
chip -- K = 113 is the last launch that is resident all at once.  A step in t / K there is the price of the second,
nearly empty round.

    python scripts/lane_vs_subsamples.py [launches per point = 20]
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
for K in (64, 96, 104, 112, 113, 114, 120, 128, 160, 192, 224, 227, 228, 256):
    v = synthetic.Visit("cfg4", det, gr, cal, n_exposures=1, E=0.983e9 * K / 128., K=K)
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
    t = {k: p[k]["ms"] / max(p[k]["launches"], 1) * 1e3 for k in ("k_prep_sub", "k_lane", "k_narrow")}
    print("K = %3d  E = %.3g   " % (K, p["electrons"] / reps) +
          "  ".join("%s %6.1f us (%.3f / sub-sample)" % (k, x, x / K) for k, x in t.items()), flush=True)
