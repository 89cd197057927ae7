#!/usr/bin/env python3
"""Turn one scripts/collect_profiles.sh run into the two files bench.py quotes, stamped with the code they
were measured on:

    python scripts/make_profile_stamps.py gpurun_out/prof_<tag> <commit> [<config>/<f32|f64>]
      -> <dir>/k_ramp_traffic.json   HBM bytes per k_ramp launch (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes,
                                      separate --pmc passes: MI355X_MICROARCH.md "HBM")
      -> <dir>/valu_issue.json       VALU issue occupancy and lane utilisation per kernel (SQ counters)

Both carry `commit`, `csrc_hash` (bench.csrc_hash(): sha256 of wayne_amd/csrc + include/wayne_hip.h) and the
kernel's average duration from the --kernel-trace --stats pass of the same collection; bench.py prints the
numbers only while csrc_hash equals the tree's.  Copy them to profiles/ to publish them.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_avg_ns(prof_dir):
    out = {}
    for f in glob.glob(os.path.join(prof_dir, "trace_split", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"].replace("void ", "").split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
    return out


def main():
    prof_dir, commit = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else "cfg4/f32"
    import bench
    stamp = {"commit": commit, "csrc_hash": bench.csrc_hash()}
    avg = kernel_avg_ns(prof_dir)
    hbm = json.load(open(os.path.join(prof_dir, "pmc_hbm.json")))
    ramp = [k for k in hbm if "k_ramp" in k and "hbm_bytes_per_launch" in hbm[k]]
    traffic = dict(stamp)
    if ramp:
        k = ramp[0]
        traffic[key] = {"kernel": k, "hbm_bytes_per_launch": hbm[k]["hbm_bytes_per_launch"]["total"],
                        "read_bytes": hbm[k]["hbm_bytes_per_launch"]["read_2xFETCH"],
                        "write_bytes": hbm[k]["hbm_bytes_per_launch"]["write"],
                        "kernel_avg_ns": avg.get(k, {}).get("avg_ns"),
                        "source": "%s/pmc_hbm.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
                                  "`bench.py --steps 4`; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per "
                                  "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts half of a coalesced stream); commit %s" % (
                                      os.path.basename(prof_dir.rstrip("/")), commit[:10])}
    json.dump(traffic, open(os.path.join(prof_dir, "k_ramp_traffic.json"), "w"), indent=1)

    sq = json.load(open(os.path.join(prof_dir, "pmc_sq.json")))
    issue = dict(stamp)
    issue["source"] = "%s/pmc_sq.json (rocprofv3 --pmc SQ counters, scripts/collect_profiles.sh), commit %s" % (
        os.path.basename(prof_dir.rstrip("/")), commit[:10])
    issue["note"] = ("a wave64 VALU instruction holds a SIMD for 4 cycles, a transcendental for 11.5 (measured: "
                     "scripts/ubench/issue_cost.hip, profiles/r02/issue_cost_ubench.txt): "
                     "issue_cycles_per_simd = (4 SQ_INSTS_VALU + 7.5 SQ_INSTS_VALU_TRANS_F32) / 1024 SIMDs; busy = "
                     "SQ_BUSY_CYCLES / 32 shader engines; valu_issue_frac = min(1, issue / busy); lane_utilisation = "
                     "SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)")
    issue["kernels"] = {}
    for k, c in sq.items():
        if "wayne::" not in k or not all(n in c for n in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_THREAD_CYCLES_VALU",
                                                           "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_TRANS_F32")):
            continue
        valu, trans = c["SQ_INSTS_VALU"]["mean"], c["SQ_INSTS_VALU_TRANS_F32"]["mean"]
        busy = c["SQ_BUSY_CYCLES"]["mean"] / 32.0
        if busy < 20000:          # tiny kernels say nothing
            continue
        cyc = (4 * valu + 7.5 * trans) / 1024.0
        issue["kernels"][k.replace("wayne::", "")] = {
            "valu_wave_instructions": round(valu), "transcendental_wave_instructions": round(trans),
            "issue_cycles_per_simd": round(cyc), "busy_cycles": round(busy),
            "valu_issue_frac": round(min(1.0, cyc / busy), 3), "model_ratio": round(cyc / busy, 3),
            "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"]["mean"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]["mean"]), 3),
            "kernel_avg_ns": avg.get(k, {}).get("avg_ns")}
    json.dump(issue, open(os.path.join(prof_dir, "valu_issue.json"), "w"), indent=1)
    print("stamped", prof_dir, stamp)


if __name__ == "__main__":
    main()
