#!/usr/bin/env python3
"""Sweep the thrower's launch geometry (workgroups per exposure, LDS tile size, margin) on cfg4."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import calibration, detector, engine, grism, synthetic  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402

cal = calibration.CalibrationSet.synthetic(11)
det = detector.WFC3_IR()
gr = grism.G141(cal)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
v = synthetic.Visit(name, det, gr, cal, n_exposures=1)
eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
ctx = eng.ctx
eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed)
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for wgs, tile, margin in itertools.product([512, 1024, 1536, 2048, 3072, 4096], [0], [16, 20, 24]):
    ctx.set_knob("throw_wgs", wgs)
    ctx.set_knob("tile_ints", tile if tile else None)
    desc = eg.build_descriptor(eng, rng_mode=mode, out_dtype=np.float32, **v.frame_kwargs(0))
    desc.thrower_margin = margin
    ctx.upload(0, desc)
    ctx.run(0)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(4):
        ctx.run(0)
    p = ctx.profile_get()
    ctx.profile_enable(False)
    print("wgs=%5d tile=%6d margin=%2d  throw=%.3f ms narrow=%.3f" % (wgs, tile, margin, p["k_throw"]["ms"] / p["k_throw"]["launches"], p["k_narrow"]["ms"] / max(p["k_narrow"]["launches"], 1)),
          flush=True)
