#!/usr/bin/env python3
"""Soak test of the pipelined path (GPU only): many exposures through VisitRunner (alternating
streams, forked thrower kernels, pinned staging and fetch buffers reused hundreds of times), then a
sample of them regenerated one at a time and compared bit for bit.

    python scripts/soak.py [config=cfg3] [n=1500]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import calibration, detector, grism, synthetic, visit  # noqa: E402
from wayne_amd.exposure_generator import ExposureGenerator  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    cal = calibration.CalibrationSet.synthetic(11)
    det = detector.WFC3_IR()
    gr = grism.G141(cal) if synthetic.CONFIGS[name]["grism"] == "G141" else grism.G102(cal)
    v = synthetic.Visit(name, det, gr, cal, n_exposures=n)
    runner = visit.VisitRunner(v, 0, frame_overrides={})
    sums = {}
    t0 = time.perf_counter()
    runner.run(range(n), on_reads=lambda i, r: sums.__setitem__(i, (float(r[-1].sum()), float(r[1].max()), r[-1][::7, ::5].copy())))
    dt = time.perf_counter() - t0
    print("%s: %d exposures delivered in %.2f s (%.0f/s)" % (name, n, dt, n / dt), flush=True)
    bad = 0
    for i in sorted(set([0, 1, 2, n // 3, n // 2, n - 2, n - 1] + list(np.random.default_rng(1).integers(0, n, 12)))):
        eg = ExposureGenerator(det, gr, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=cal, seed=v.seed, exposure_index=int(i))
        reads = np.stack([r[0] for r in eg.scanning_frame(out_dtype=np.float32, **v.frame_kwargs(int(i))).reads])
        ok = (float(reads[-1].sum()) == sums[i][0] and float(reads[1].max()) == sums[i][1]
              and np.array_equal(reads[-1][::7, ::5], sums[i][2]))
        bad += 0 if ok else 1
        if not ok:
            print("exposure %d differs between the pipelined and the one-at-a-time path" % i)
    assert len(sums) == n and bad == 0, "%d of the sampled exposures differ" % bad
    assert all(np.isfinite(s[0]) for s in sums.values())
    print("soak ok: %d exposures, sampled ones bit-identical to one-at-a-time generation" % n)


if __name__ == "__main__":
    main()
