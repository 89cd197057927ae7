#!/usr/bin/env python3
"""Exposures/s of every BASELINE.json configuration on one MI355X: device-complete (reads left in HBM; one
stream, and alternating over two) and end to end (host descriptor -> upload -> kernels -> reads in pinned host
memory, light curves on the device: VisitRunner)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402   (timed_pass / median_of_passes: one way of timing a pass for both scripts)
from wayne_amd import calibration, detector, engine, grism, synthetic, visit  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402

cal = calibration.CalibrationSet.synthetic(11)
det = detector.WFC3_IR()
out = {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2     # 2 = split thrower (default), 1 = every electron
for name in ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg5_g102"]:
    gr = grism.G141(cal) if synthetic.CONFIGS[name]["grism"] == "G141" else grism.G102(cal)
    v = synthetic.Visit(name, det, gr, cal, n_exposures=n + 2)
    eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    for j in range(n + 2):
        eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed, exposure_index=j)
        ctx.upload(2 * j, eg.build_descriptor(eng, rng_mode=mode, out_dtype=np.float32, **v.frame_kwargs(j)))
    # (median of three passes of n after 4 warm-up exposures: bench.median_of_passes, as bench.py's two_streams)
    rate_one, _ = bench.median_of_passes(ctx, lambda j: 2 * (j % (n + 2)), n, reps=3, warmup=4)
    dt = n / rate_one
    # kernel times from a second pass (HIP events around every kernel cost a few per cent of throughput)
    ctx.profile_enable(True)
    ctx.profile_reset()
    for j in range(2, n + 2):
        ctx.run(2 * j)
    p = ctx.profile_get()
    ctx.profile_enable(False)
    # the same exposures alternating over the context's two streams
    for j in range(n + 2):
        eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed, exposure_index=j)
        ctx.upload(j, eg.build_descriptor(eng, rng_mode=mode, out_dtype=np.float32, **v.frame_kwargs(j)))
    rate_two, _ = bench.median_of_passes(ctx, lambda j: j % (n + 2), n, reps=3, warmup=4)
    dt_two = n / rate_two
    # end to end: the visit runner (host prep + upload + kernels + copy to pinned host memory), device light curves
    runner = visit.VisitRunner(v, 0, frame_overrides={}, device_lc=True)
    runner.rng_mode = mode
    runner.run(list(range(4)))
    t1 = time.perf_counter()
    runner.run([j % (n + 2) for j in range(3 * n)])
    dt2 = (time.perf_counter() - t1) / 3
    out[name] = {"device_complete_exp_s": n / dt, "two_streams_exp_s": n / dt_two, "end_to_end_exp_s": n / dt2,
                 "electrons_per_exposure": p["electrons"] / n,
                 "ms": {k: p[k]["ms"] / n for k in p if k != "electrons"}, "K": v.K, "W": int(np.sum((v.wl >= gr.wl_limits[0]) & (v.wl <= gr.wl_limits[1]))),
                 "frame": eng.N, "NSAMP": v.NSAMP}
    print(name, json.dumps(out[name]), flush=True)
    engine.close_all()
print(json.dumps(out))
