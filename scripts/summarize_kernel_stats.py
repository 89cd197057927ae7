#!/usr/bin/env python3
"""rocprofv3's <pid>_kernel_stats.csv -> a small JSON keyed by kernel family (what DESIGN.md quotes):

    python scripts/summarize_kernel_stats.py profiles/r06/kernel_stats.csv > profiles/r06/kernel_stats.json

{"k_lane": {"name": full instantiation, "calls": n, "avg_us": ..., "percent": ...}, ..., "wayne_sum_us": the sum of the
averages of the wayne:: kernels = the device time of one exposure on one stream}."""
import csv
import json
import sys


def main():
    out, total = {}, 0.0
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Name"].replace("void ", "")
        if "wayne::" not in name:
            continue
        short = name.split("wayne::")[1].split("<")[0].split("(")[0]
        full = name.split("(")[0].replace("wayne::", "")
        if short in out:            # two instantiations of one family in a run: keep the one with more calls
            if int(r["Calls"]) <= out[short]["calls"]:
                continue
            total -= out[short]["avg_us"]
        out[short] = {"name": full, "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                      "percent": float(r["Percentage"])}
        total += out[short]["avg_us"]
    out["wayne_sum_us"] = total
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
