#!/usr/bin/env python3
"""sha256 of the reads of a few exposures of a configuration, in the production mode and per electron (GPU only): for
showing that two builds of the library produce the SAME frames bit for bit.

    WAYNE_HIP_LIB=ab/A.so python scripts/frame_hash.py cfg4 3;  WAYNE_HIP_LIB=ab/B.so python scripts/frame_hash.py cfg4 3
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from wayne_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
v = helpers.make_visit(name, n_exposures=n)
for mode, label in ((_lib.RNG_SPLIT, "split"), (_lib.RNG_PHILOX, "philox")):
    h = hashlib.sha256()
    for i in range(n):
        pg = helpers.product_generator(v, i)
        reads = np.stack([r[0] for r in pg.scanning_frame(rng_mode=mode, **v.frame_kwargs(i)).reads]) if v.scan_speed else \
            np.stack([r[0] for r in pg.staring_frame(rng_mode=mode, **{k: x for k, x in v.frame_kwargs(i).items()
                                                                        if k not in ("scan_speed", "sample_rate", "ssv_generator")}).reads])
        h.update(reads.tobytes())
    print("%s %s %d exposures %s" % (name, label, n, h.hexdigest()[:32]), flush=True)
