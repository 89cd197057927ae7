#!/usr/bin/env python3
"""Ablation timing of the exposure kernels on the metric's config (GPU only).

    python scripts/ablate.py [cfg4] [reps]

Runs the same exposure with stages switched off by flag and prints the HIP-event
time of every kernel: which stage of k_ramp / k_throw costs what.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import calibration, detector, engine, grism, synthetic  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cal = calibration.CalibrationSet.synthetic(11)
    det = detector.WFC3_IR()
    gr = grism.G141(cal)
    v = synthetic.Visit(name, det, gr, cal, n_exposures=1)
    eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    variants = {
        "all_on": {},
        "no_sky": dict(sky_background=0.0),
        "no_nonlin": dict(add_non_linear=False),
        "no_dark": dict(add_dark=False),
        "no_readnoise_no_dark": dict(add_dark=False, add_read_noise=False),
        "no_cosmic": dict(cosmic_rate=None),
        "no_flat": dict(add_flat=False),
        "no_stellar_noise": dict(add_stellar_noise=False),
        "bare": dict(sky_background=0.0, add_non_linear=False, add_dark=False, add_read_noise=False,
                     cosmic_rate=None, add_gain_variations=False, clip_values_det_limits=False),
    }
    out = {}
    for vn, over in variants.items():
        eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed)
        desc = eg.build_descriptor(eng, out_dtype=np.float32, **v.frame_kwargs(0, **over))
        ctx.upload(0, desc)
        ctx.run(0)
        ctx.synchronize()
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(reps):
            ctx.run(0)
        p = ctx.profile_get()
        ctx.profile_enable(False)
        out[vn] = {k: round(p[k]["ms"] / max(p[k]["launches"], 1), 4) for k in p if k != "electrons"}
        print("%-22s" % vn, " ".join("%s=%.3f" % (k[2:], t) for k, t in out[vn].items() if k != "other"), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
