"""Print how closely the device's split-mode thrower follows oracle/split_oracle.c on the same counters."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden_psf
from oracle import clib
from wayne_amd import _lib

ctx = _lib.default_context()
for name, scale in [("s256_t4", 1), ("s256_t4", 60), ("s1014_t4", 8), ("edge_low", 20), ("ratio_01", 5)]:
    k = load_golden_psf(name)
    n = k["nr"]
    counts = (k["counts"].astype(np.int64) * scale).astype(np.int32)
    want = clib.psf_split_oracle(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"], n, 1963, 0, 0)
    got = ctx.psf_apply(counts, k["x"], k["y"], k["ratio"], k["sl"], k["sh"], n, n, 1963, rng_mode=_lib.RNG_SPLIT)
    moved = int(np.abs(got.astype(np.int64) - want).sum()) // 2
    print("%-10s x%-3d electrons %10d  totals %d/%d  moved %d (%.2e)" % (name, scale, counts.sum(), got.sum(), want.sum(), moved, moved / max(want.sum(), 1)))
