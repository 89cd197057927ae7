#!/usr/bin/env python3
"""Visit-level parity in the units a user cares about: transit depths in ppm (GPU only).

    python scripts/visit_science.py [--cfg3 1024] [--cfg4 160] [--replay-every 8] [--out gpurun_out/visit_science.json]

Generates a cfg3-shaped (256 x 256, NSAMP 15, 64 sub-samples, 4e8 e-) and a cfg4-shaped (1014 x 1014, NSAMP 16, 128
sub-samples, 1e9 e-) visit with SURVEY.md 8(d)'s depth spectrum injected, in the production mode (split thrower,
float32 reads), with float64 reads, per electron with float64 reads, and -- every n-th exposure -- in the bit-exact
replay mode; extracts 20 spectral light curves per mode as an observer would and fits the depths
(tests/visit_science.py holds the extraction and the fits; tests/test_visit_science_gpu.py asserts on a shorter run of
the same).  The report replaces the argued bias column of the production thrower's budget by measured numbers.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import visit_science as vs  # noqa: E402


def run(name, n, replay_every, modes=("production", "split_f64", "per_electron")):
    sv = vs.ScienceVisit(name, n)
    tables, subset, seconds = {}, {}, {}
    for mode in modes:
        t = time.perf_counter()
        tables[mode] = vs.generate(sv, mode)
        seconds[mode] = round(time.perf_counter() - t, 2)
        print("%s %s: %d exposures in %.1f s" % (name, mode, n, seconds[mode]), flush=True)
    if replay_every:
        idx = np.arange(0, n, replay_every)
        t = time.perf_counter()
        tables["replay"] = vs.generate(sv, "replay", idx)
        subset["replay"] = idx
        seconds["replay"] = round(time.perf_counter() - t, 2)
        print("%s replay: %d exposures in %.1f s" % (name, len(idx), seconds["replay"]), flush=True)
    rep = vs.analyse(sv, tables, subset)
    rep["seconds_generate_and_extract"] = seconds
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg3", type=int, default=1024)
    ap.add_argument("--cfg4", type=int, default=160)
    ap.add_argument("--cfg2", type=int, default=0, help="staring mode, full array (BASELINE configs[1]: 100 exposures)")
    ap.add_argument("--cfg5-g102", type=int, default=0, help="the G102 grism, cfg4's shape with SSV")
    ap.add_argument("--replay-every", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "visit_science.json"))
    a = ap.parse_args()
    out = {}
    if a.cfg3:
        out["cfg3"] = run("cfg3", a.cfg3, a.replay_every)
    if a.cfg4:
        out["cfg4"] = run("cfg4", a.cfg4, a.replay_every)
    if a.cfg2:
        out["cfg2"] = run("cfg2", a.cfg2, a.replay_every)
    if a.cfg5_g102:
        out["cfg5_g102"] = run("cfg5_g102", a.cfg5_g102, a.replay_every)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name, rep in out.items():
        print("==", name, rep["n_exposures"], "exposures")
        for mode, m in rep["modes"].items():
            r = m["ramp"]
            print("  %-13s n=%4d  white rec-inj %+8.2f +- %.2f ppm   channel chi2 %.1f / %d   rms/photon %.2f" % (
                mode, m["n"], r["white_recovered_minus_injected_ppm"], r["white_sigma_ppm"], r["chi2"], r["dof"],
                float(np.mean(r["residual_rms_over_photon_noise"]))))
        for pair, m in rep["paired"].items():
            r = m["ramp"]
            print("  %-28s n=%4d  white %+7.3f +- %.3f ppm   channels chi2 %.1f / %d (sigma %.1f ppm)   x-phase chi2 %.1f  y-phase chi2 %.1f / 2" % (
                pair, m["n"], r["white_depth_difference_ppm"], r["white_sigma_ppm"], r["chi2"], r["dof"],
                float(np.mean(r["sigma_ppm"])), r["flux_ratio_vs_x_phase_ppm"]["chi2"], r["flux_ratio_vs_y_phase_ppm"]["chi2"]))


if __name__ == "__main__":
    main()
