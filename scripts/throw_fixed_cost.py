#!/usr/bin/env python3
"""k_throw / k_narrow time against the number of electrons at fixed geometry (GPU only):
the intercept is the per-exposure fixed cost (tile zeroing + flush), the slope the per-electron cost.

    python scripts/throw_fixed_cost.py [cfg4] [rng_mode]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import calibration, detector, engine, grism, synthetic  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cal = calibration.CalibrationSet.synthetic(11)
    det = detector.WFC3_IR()
    gr = grism.G141(cal)
    v = synthetic.Visit(name, det, gr, cal, n_exposures=1)
    eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    for flat in (True, False):
        for f in (0.03, 0.125, 0.25, 0.5, 1.0, 1.5):
            eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed)
            kw = v.frame_kwargs(0, add_flat=flat)
            kw["scale_factor"] = kw["scale_factor"] * f
            desc = eg.build_descriptor(eng, rng_mode=mode, out_dtype=np.float32, **kw)
            ctx.upload(0, desc)
            ctx.run(0)
            ctx.synchronize()
            ctx.profile_enable(True)
            ctx.profile_reset()
            for _ in range(5):
                ctx.run(0)
            p = ctx.profile_get()
            ctx.profile_enable(False)
            t = {k: p[k]["ms"] / max(p[k]["launches"], 1) for k in p if k != "electrons"}
            print("flat=%d scale %.3f electrons %.3e  throw %.4f narrow %.4f prep %.4f ramp %.4f" % (
                flat, f, p["electrons"] / 5, t["k_throw"], t["k_narrow"], t["k_prep_sub"], t["k_ramp"]), flush=True)


if __name__ == "__main__":
    main()
