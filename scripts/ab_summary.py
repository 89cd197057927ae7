#!/usr/bin/env python3
"""Lines of scripts/ab_run.sh (`<lib>: <cfg> <rate> exposures/s (best of 5 x 40)  {kernel: ms, ...}`) -> JSON:

    python scripts/ab_summary.py <before lib> <after lib> file.txt [file.txt ...] > profiles/r06/ab_bin_local.json

{cfg: {"before": mean rate of <before lib>, "after": ..., "change_percent": ..., "kernels_before_ms": {...}, ...}}
"""
import ast
import json
import re
import sys

LINE = re.compile(r"^(\S+): (\S+) (\d+) exposures/s \(best of [^)]*\)\s+(\{.*\})\s*$")


def main():
    before, after, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    rows = {}
    for f in files:
        for line in open(f):
            m = LINE.match(line)
            if not m or m.group(1) not in (before, after):
                continue
            lib, cfg, rate, kern = m.group(1), m.group(2), float(m.group(3)), ast.literal_eval(m.group(4))
            rows.setdefault(cfg, {}).setdefault(lib, []).append((rate, kern))
    out = {"libraries": {"before": before, "after": after}, "sources": files,
           "note": "two libraries alternating inside ONE gpurun call (boxes differ by a few per cent, runs on one box by ~0.5 %)"}
    for cfg, by in sorted(rows.items()):
        if before not in by or after not in by:
            continue
        def mean(lib, key=None):
            v = [r if key is None else k[key] for r, k in by[lib]]
            return sum(v) / len(v)
        kernels = sorted(set(by[before][0][1]) & set(by[after][0][1]))
        out[cfg] = {"before": mean(before), "after": mean(after), "runs_each": len(by[before]),
                    "change_percent": 100.0 * (mean(after) / mean(before) - 1.0),
                    "kernels_before_ms": {k: round(mean(before, k), 4) for k in kernels},
                    "kernels_after_ms": {k: round(mean(after, k), 4) for k in kernels}}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
