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
    issue["note"] = (
        "valu_busy_frac = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles): SQ_ACTIVE_INST_VALU counts the QUAD-cycles a "
        "SIMD spends executing vector instructions (MI355X_MICROARCH.md: 'SQ_ACTIVE_INST_* count quad-cycles'; it equals "
        "SQ_INSTS_VALU + SQ_INSTS_VALU_TRANS_F32 on these kernels: one quad-cycle per wave64 instruction, two per "
        "transcendental); cycles = the kernel's average duration (rocprofv3 --kernel-trace --stats of the same "
        "collection) x 2.4 GHz, the chip's maximum clock -- the clock under load is at most that, so the figure is a "
        "LOWER bound of the fraction of SIMD cycles with a vector instruction executing.  Counted, not modelled, not "
        "clamped.  valu_busy_frac_gui is the same over GRBM_GUI_ACTIVE / 8 XCDs of the dispatch, which on dispatches "
        "this short includes cycles around the kernel (the guide: 'reads high on dispatches shorter than about 0.3 ms'; "
        "gui_clock_ghz = those cycles / the duration comes out above 2.4) and so reads lower still.  (The SQ's own "
        "busy counters are not wall-time denominators: SQ_BUSY_CYCLES / 32 shader engines and SQ_BUSY_CU_CYCLES / 256 "
        "CUs count 0.88-0.93 of duration x 2.4 GHz on the big kernels -- idle stretches of an engine or CU are not in "
        "them -- and the vector quad-cycles per SIMD come to 0.99-1.04 of them.)  lane_utilisation = "
        "SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)")
    issue["kernels"] = {}
    for k, c in sq.items():
        if "wayne::" not in k or not all(n in c for n in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_THREAD_CYCLES_VALU",
                                                           "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_TRANS_F32")):
            continue
        valu, trans = c["SQ_INSTS_VALU"]["mean"], c["SQ_INSTS_VALU_TRANS_F32"]["mean"]
        active = c["SQ_ACTIVE_INST_VALU"]["mean"]
        ns = avg.get(k, {}).get("avg_ns")
        if c["SQ_BUSY_CYCLES"]["mean"] / 32.0 < 20000 or not ns:          # tiny kernels say nothing
            continue
        cycles = ns * 2.4
        gui = c["GRBM_GUI_ACTIVE"]["mean"] / 8.0 if "GRBM_GUI_ACTIVE" in c else None
        issue["kernels"][k.replace("wayne::", "")] = {
            "valu_wave_instructions": round(valu), "transcendental_wave_instructions": round(trans),
            "valu_active_quad_cycles": round(active),
            "kernel_avg_ns": ns, "cycles_at_2p4_ghz": round(cycles),
            "valu_busy_frac": round(4.0 * active / 1024.0 / cycles, 3),
            "valu_busy_frac_gui": round(4.0 * active / 1024.0 / gui, 3) if gui else None,
            "gui_clock_ghz": round(gui / ns, 3) if gui else None,
            "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"]["mean"] / (64.0 * active), 3)}
    json.dump(issue, open(os.path.join(prof_dir, "valu_issue.json"), "w"), indent=1)
    print("stamped", prof_dir, stamp)


if __name__ == "__main__":
    main()
