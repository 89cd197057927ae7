#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel.

    python scripts/summarize_pmc.py gpurun_out/prof_r1/pmc_fetch gpurun_out/prof_r1/pmc_write > profiles/rNN/pmc.json

HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a
coalesced stream, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import sys


def main(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in sorted(agg.items()):
        out[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            rd = 2 * out[k]["FETCH_SIZE"]["mean"] * 1024
            wr = out[k]["WRITE_SIZE"]["mean"] * 1024
            out[k]["hbm_bytes_per_launch"] = {"read_2xFETCH": rd, "write": wr, "total": rd + wr}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
