#!/usr/bin/env python3
"""What bounds the delivered rate?  On the GPU box:
    python scripts/probe_pipeline.py
prints (a) the device-to-host rate of the reads alone (fetch_async + wait on finished exposures, no kernels),
(b) the VisitRunner pipeline on resident descriptors at several depths, (c) kernels only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WAYNE_STREAMS", "2")

from wayne_amd import calibration, detector, engine, grism, synthetic, visit as wv   # noqa: E402


def main():
    cal = calibration.CalibrationSet.synthetic(11)
    det = detector.WFC3_IR()
    gr = grism.G141(cal)
    v = synthetic.Visit("cfg4", det, gr, cal, n_exposures=8)
    eng = engine.get_engine(0, gr, det, cal, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    runner = wv.VisitRunner(v)
    for s in range(8):
        ctx.upload(s, runner.descriptor(s, eng))
        ctx.run(s)
        ctx.fetch_async(s)
        ctx.wait(s)
    mb = (eng.R + 1) * eng.S * eng.S * 4 / 1e6
    n = 60
    for depth in (1, 2, 4):
        ctx.synchronize()
        t = time.perf_counter()
        pend = []
        for j in range(n):
            s = j % depth
            if len(pend) == depth:
                ctx.wait(pend.pop(0))
            ctx.fetch_async(s)
            pend.append(s)
        for s in pend:
            ctx.wait(s)
        dt = time.perf_counter() - t
        print("copy only, %d in flight: %.1f /s = %.1f GB/s" % (depth, n / dt, n * mb / dt / 1e3), flush=True)
    # copies of slots 0..1 while the kernels of slots 4..7 keep the GPU busy (never waited for inside the loop)
    ctx.synchronize()
    t = time.perf_counter()
    pend = []
    for j in range(n):
        s = j % 2
        if len(pend) == 2:
            ctx.wait(pend.pop(0))
        ctx.fetch_async(s)
        pend.append(s)
        ctx.run(4 + j % 4)
    for s in pend:
        ctx.wait(s)
    dt = time.perf_counter() - t
    ctx.synchronize()
    dt2 = time.perf_counter() - t
    print("copies beside running kernels: %.1f /s = %.1f GB/s (kernels drained after %.3f s, copies after %.3f s)" % (
        n / dt, n * mb / dt / 1e3, dt2, dt), flush=True)
    for depth in (2, 3, 4, 6, 8):
        runner.DEPTH = depth
        runner.run_resident(depth)
        ctx.synchronize()
        t = time.perf_counter()
        runner.run_resident(n)
        dt = time.perf_counter() - t
        print("pipeline depth %d: %.1f /s = %.1f GB/s" % (depth, n / dt, n * mb / dt / 1e3), flush=True)
    ctx.synchronize()
    t = time.perf_counter()
    for j in range(n):
        ctx.run(j % 8)
    ctx.synchronize()
    dt = time.perf_counter() - t
    print("kernels only, two streams: %.1f /s" % (n / dt), flush=True)


if __name__ == "__main__":
    main()
