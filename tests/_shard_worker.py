"""Child process of tests/test_sharding.py (GPU): generate this rank's round-robin shard of a tiny visit
on device 0 and save the frames.  Usage: _shard_worker.py RANK WORLD N_EXPOSURES OUT_DIR"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, n_exp, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import helpers
    from wayne_amd import visit as wv
    v = helpers.make_visit("tiny", n_exposures=n_exp)
    runner = wv.VisitRunner(v, device=0, out_dtype=np.float64)
    frames = runner.run(wv.shard(n_exp, rank, world), keep=True)
    np.savez(os.path.join(out, "rank%d.npz" % rank), **{"e%d" % i: f for i, f in frames.items()})


if __name__ == "__main__":
    main()
