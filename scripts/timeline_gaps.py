#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace run (csv): python scripts/timeline_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list)
gap = defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev is not None:
        gap[prev[0] + " -> " + name].append(s - prev[1])
    prev = (name, e)
med = lambda v: sorted(v)[len(v) // 2]
print("kernel durations (median ns, n)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s %8d  n=%d" % (k[:60], med(v), len(v)))
print("gaps end->start (median ns, n)")
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:14]:
    print("  %-90s %8d  n=%d" % (k[:90], med(v), len(v)))
