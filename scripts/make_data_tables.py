#!/usr/bin/env python3
"""Convert the reference's small DATA tables into this package's own formats.

Runs in the build container only (reads /root/reference/wayne/data).  Data,
not source: the WFC3-IR mode timing table (HST Phase II Proposal Instructions
sec. 13.3.6, via wayne/data/wfc3_ir_mode_exptime.csv), the dark-file-per-mode
table (wfc3_ir_mode_calb.csv) and the 266x266 initial bias frame
(wfc3_ir_initial_bias_256.fits).

  wayne_amd/data/wfc3_ir_modes.json   {"exptime": {SUBARRAY: {SAMPSEQ: [t_1..t_n]}},
                                        "dark_file": {SUBARRAY: {SAMPSEQ: name}}}
  wayne_amd/data/wfc3_ir_initial_bias_256.npy   float64 (266, 266)
"""
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import fitsio  # noqa: E402

REF = "/root/reference/wayne/data"
OUT = os.path.join(ROOT, "wayne_amd", "data")


def main():
    exptime = {}
    with open(os.path.join(REF, "wfc3_ir_mode_exptime.csv"), newline="") as f:
        rows = list(csv.reader(f))[2:]
    n = 0
    for sub, seq, num, t in rows:
        times = exptime.setdefault(sub, {}).setdefault(seq, [])
        assert int(num) == len(times) + 1, (sub, seq, num)   # SAMPNUM ascending from 1
        times.append(float(t.replace(",", "")))
        n += 1
    dark = {}
    with open(os.path.join(REF, "wfc3_ir_mode_calb.csv"), newline=None) as f:
        text = f.read().replace("\r", "\n")
    rows = [r for r in csv.reader(text.splitlines()) if r][2:]
    for sub, seq, name in rows:
        dark.setdefault(sub, {})[seq] = name
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "wfc3_ir_modes.json"), "w") as f:
        json.dump({"exptime": exptime, "dark_file": dark, "rows": {"exptime": n, "dark_file": len(rows)}},
                  f, indent=0, sort_keys=True)
    bias = fitsio.read(os.path.join(REF, "wfc3_ir_initial_bias_256.fits"))[1].data
    assert bias.shape == (266, 266)
    np.save(os.path.join(OUT, "wfc3_ir_initial_bias_256.npy"), bias.astype(np.float64))
    print("exptime rows", n, "dark rows", len(rows), "bias", bias.shape, bias.dtype, float(bias.mean()))


if __name__ == "__main__":
    main()
