#!/usr/bin/env python3
"""Where the HOST spends its time per exposure in the end-to-end pipeline (GPU box): cProfile over
VisitRunner.run of a configuration, device light curves, reads delivered to pinned host memory.

    python scripts/profile_host_path.py [config=cfg1] [exposures=300]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from wayne_amd import visit as wv  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
v = helpers.make_visit(name, n_exposures=n)
runner = wv.VisitRunner(v, device=0, device_lc=True)
runner.run(range(8))
t0 = time.perf_counter()
runner.run(range(n))
dt = time.perf_counter() - t0
print("%s: %d exposures end to end in %.3f s = %.0f exposures/s (%.3f ms each)" % (name, n, dt, n / dt, dt / n * 1e3))
pr = cProfile.Profile()
pr.enable()
runner.run(range(n))
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
from wayne_amd import engine  # noqa: E402
os.environ["WAYNE_UPLOAD_TIMING"] = "1"
engine.close_all()
