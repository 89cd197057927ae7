#!/usr/bin/env python3
"""Per-kernel times and exposures/s of one configuration, for A/B comparisons of two builds of libwayne_hip.so INSIDE
one gpurun call (boxes differ by a few per cent, runs on one box by ~0.5 %):

    cp ab/A.so wayne_amd/libwayne_hip.so && python scripts/ab_kernels.py cfg5 && cp ab/B.so wayne_amd/libwayne_hip.so && ...

    python scripts/ab_kernels.py [config=cfg5] [exposures=40] [rng_mode=2 (0 replay, 1 every electron, 2 split)]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import calibration, detector, engine, grism, synthetic  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cal = calibration.CalibrationSet.synthetic(11)
det = detector.WFC3_IR()
gr = grism.G141(cal) if synthetic.CONFIGS[name]["grism"] == "G141" else grism.G102(cal)
v = synthetic.Visit(name, det, gr, cal, n_exposures=12)
eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
ctx = eng.ctx
slots = 10
for j in range(slots):
    eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed, exposure_index=j)
    ctx.upload(2 * j, eg.build_descriptor(eng, rng_mode=mode, threads=2, out_dtype=np.float32, **v.frame_kwargs(j)))
for j in range(slots):
    ctx.run(2 * j)
ctx.synchronize()
best = 0.0
for rep in range(5):
    t0 = time.perf_counter()
    for j in range(n):
        ctx.run(2 * (j % slots))
    ctx.synchronize()
    best = max(best, n / (time.perf_counter() - t0))
ctx.profile_enable(True)
ctx.profile_reset()
for j in range(n):
    ctx.run(2 * (j % slots))
p = ctx.profile_get()
ctx.profile_enable(False)
ms = {k: round(x["ms"] / max(x["launches"], 1), 4) for k, x in p.items() if isinstance(x, dict) and x["launches"]}
print("%s %.0f exposures/s (best of 5 x %d)  %s" % (name, best, n, ms))
