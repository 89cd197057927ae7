#!/usr/bin/env python3
"""Offline fuzz of the product's host-side launch planner under AddressSanitizer + UBSan (CPU only; the suite's
tests/test_host_plan.py holds the fixed-seed property tests, this is the longer search):

    python scripts/fuzz_host_plan.py [first seed = 0] [last seed = 12]

Per seed: 150 random descriptors (a third of them hostile, some with NaN / negative / infinite durations and scale
factors on top), 40 sky plans and 40 thrower calls made of junk values -- all through tests/native/plan_harness (the
SAME host_plan.h the library includes).  A sanitizer report stops the run; the box property (every position the
oracle's trace computes, +- 6.9 sigma, inside box[read], or "load everything") is checked on every clean descriptor.
Round 5: seeds 0-11 (1800 plans -- 160 of them with a sensitivity table the ABI refuses -- 480 sky plans, 480 thrower
calls): no report, 0 violations.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import plan_harness as ph  # noqa: E402
import test_host_plan as t  # noqa: E402

ph.build()
bad_total = 0
t0 = time.time()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for seed in range(first, last):
    rng = np.random.default_rng(1000 + seed)
    cases = [t.random_descriptor(rng, hostile=(i % 3 == 0)) for i in range(150)]
    # extra hostility: tiny / huge K, W = 2, durations NaN / negative, scale NaN
    for d in cases[::7]:
        d["dur"][rng.integers(0, d["dur"].size)] = rng.choice([np.nan, -5.0, np.inf, 0.0])
        d["scale"] = float(rng.choice([np.nan, 0.0, -1.0, 1e300, 1.0]))
    # sensitivity tables the ABI refuses (the planner must still stay inside them): junk entries, steps back, shuffles
    for d in cases[3::11]:
        w, v = d["ga"]["sens_wl"].copy(), d["ga"]["sens_val"].copy()
        how = rng.choice(["junk", "reverse", "shuffle", "short"])
        if how == "junk":
            for a in (w, v):
                a[rng.integers(0, a.size, 3)] = rng.choice([np.nan, np.inf, -np.inf, 1e300, -1.0], size=3)
        elif how == "reverse":
            w = w[::-1].copy()
        elif how == "shuffle":
            w = rng.permutation(w)
        else:
            w, v = w[:2].copy(), v[:2].copy()
            w[rng.integers(0, 2)] = np.nan
        d["ga"] = dict(d["ga"], sens_wl=w, sens_val=v)
    b = ph.Batch()
    idx = [t.add_plan(b, d) for d in cases]
    # sky ops with random junk
    for _ in range(40):
        n = int(rng.choice([0, 1, 3, 500]))
        sky = np.sort(np.abs(rng.normal(1, 0.3, n)).astype(np.float32))
        if n and rng.random() < 0.3:
            sky[rng.integers(0, n)] = np.float32(rng.choice([np.nan, np.inf, 0.0, 1e30]))
        dt = rng.choice([np.nan, 0.0, -1.0, 2.9, 10.0, 1e9, np.inf], size=int(rng.integers(0, 20)))
        b.sky(float(rng.choice([0.0, 5.0, np.nan, 1e6, -2.0, 1e-30])), dt, sky, has_sky=bool(rng.random() < 0.8))
    for _ in range(40):
        n = int(rng.choice([0, 1, 64, 700]))
        counts = rng.integers(0, 2 ** 31 - 1, n) // int(rng.choice([1, 1000, 10 ** 6, 10 ** 9]))
        arrs = [rng.choice([np.nan, np.inf, -np.inf, 0.0, 0.2, 0.7, 1e300, -1e300, 500.0], size=n) for _ in range(4)]
        b.psf(counts, arrs[0], arrs[1], arrs[2], arrs[3], int(rng.choice([1, 64, 1014])), int(rng.choice([0, 1, 2])),
              int(rng.choice([1, 4, 2 ** 20])), int(rng.choice([0, 30, 10 ** 6])))
    out = b.run()
    for d, i in zip(cases, idx):
        r = out[i]
        for key, n in (("chunk_order", r["n_chunks"]), ("lane_order", r["n_lane_chunks"])):
            assert sorted(r[key][:n].tolist()) == list(range(n)), (seed, key)
        if r["use_box"]:
            if d["poisoned"]:
                bad_total += 1; print("poisoned but boxed", seed)
            elif np.isfinite(d["wl"]).all():
                nb = t.positions_outside_box(d, r)
                if nb: bad_total += 1; print("outside box", seed, nb)
    print("seed", seed, "ok", round(time.time() - t0, 1), "s", flush=True)
print("violations:", bad_total)
