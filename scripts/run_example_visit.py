#!/usr/bin/env python3
"""End-to-end timing of an example-shaped visit through the CLI entry point (GPU only).

    python scripts/run_example_visit.py [n_exposures=121] [outdir] [ranks]

`ranks` > 1: `python -m wayne_amd.run_visit -p ... --gpus 1 --ranks-per-gpu ranks` as a child process, its rank
processes sharing device 0 -- on a small sub-array the visit is bound by one interpreter's lock (descriptor building + FITS
headers), not by the GPU, and several ranks on one GPU are the way past it.

Writes a parameter file with the settings of the reference's example visit
(examples/hd209458b_12181_simulation_parameters.yml: G141 spatial scan, SUBARRAY 256,
SPARS10 NSAMP 5, 10 ms sampling -> K = 2233 sub-samples per exposure, every detector
effect on, SSV sine, cosmic rays, visit trend) plus synthetic per-exposure data files
(start times over 5 orbits, reference positions, sky levels, a 15000-row planet
spectrum; black-body star), runs `wayne_amd.run_visit` on it and reports wall time:
YAML -> light curves -> exposures on the GPU -> NNNN_raw.fits on disk.
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import run_visit  # noqa: E402

PARAMS = """
general:
  oec_location: False
  outdir: 'simulated'
  seed: 1963
  threads: 4
target:
  name: 'HD 209458 b'
  planet_spectrum_file: 'planet_spectrum.dat'
  rebin_resolution: false
  stellar_spectrum_file: false
  star_temperature: 6100.0
  flux_scale: 2.8829687E-20
  period: 3.524746
  sma: 0.047309
  stellar_radius: 1.155
  inclination: 86.71
  eccentricity: 0.0
  periastron: 0.0
  transit_time: 2456196.28836
  ldcoeffs: [0.800627, -0.757066, 0.897268, -0.384804]
observation:
  detector: 'WFC3IR'
  grism: 'G141'
  x_ref: 'xref.txt'
  y_ref: 'yref.txt'
  NSAMP: 5
  SAMPSEQ: 'SPARS10'
  SUBARRAY: 256
  start_JD: False
  exp_start_times: 'jd.txt'
  num_orbits: 5
  sample_rate: 10
  spatial_scan: True
  scan_speed: 7.4325
  ssv_type: sine
  ssv_coeffs: [1.5, 1.1, 0]
  x_shifts: 0
  x_jitter: 0.025
  y_shifts: 0
  y_jitter: 0.000000000000001
  noise_mean: False
  noise_std: False
  add_dark: True
  add_flat: True
  add_gain_variations: True
  add_non_linear: True
  add_read_noise: True
  add_initial_bias: True
  add_stellar_noise: True
  sky_background: 'sky.txt'
  cosmic_rate: 11
  clip_values_det_limits: True
trends:
  visit_trend_coeffs: [0.005, 0.0011, 400, 2456196.28836]
"""


def write_inputs(work, n):
    rng = np.random.default_rng(12181)
    # 5 HST orbits of 96 min, ~50 min visible each, one exposure every ~2 min, centred on mid-transit
    per_orbit = (n + 4) // 5
    t0 = 2456196.28836 - 2.5 * 96.0 / 1440.0
    jd = []
    for o in range(5):
        for j in range(per_orbit):
            jd.append(t0 + o * 96.0 / 1440.0 + j * 2.0 / 1440.0)
    jd = np.array(jd[:n])
    np.savetxt(os.path.join(work, "jd.txt"), jd, fmt="%.8f")
    np.savetxt(os.path.join(work, "xref.txt"), 460.0 + np.cumsum(rng.normal(0, 0.003, n)), fmt="%.4f")
    np.savetxt(os.path.join(work, "yref.txt"), 400.0 + np.cumsum(rng.normal(0, 0.003, n)), fmt="%.4f")
    np.savetxt(os.path.join(work, "sky.txt"), 1.0 + 0.3 * np.sin(np.linspace(0, 9, n)) ** 2, fmt="%.3f")
    wl = np.linspace(0.5, 2.0, 15000)
    depth = 0.0146 + 2.5e-4 * np.exp(-0.5 * ((wl - 1.4) / 0.08) ** 2)          # a water-band-like bump
    np.savetxt(os.path.join(work, "planet_spectrum.dat"), np.column_stack([wl, depth]), fmt="%.10f")
    with open(os.path.join(work, "params.yml"), "w") as f:
        f.write(PARAMS)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 121
    work = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="wayne_example_")
    os.makedirs(work, exist_ok=True)
    write_inputs(work, n)
    profile = bool(os.environ.get("WAYNE_PROFILE"))     # per-kernel HIP-event times of the whole visit
    if profile:
        from wayne_amd import engine
        make = engine.Engine.__init__

        def make_and_profile(self, *a, **k):
            make(self, *a, **k)
            self.ctx.profile_enable(True)
        engine.Engine.__init__ = make_and_profile
    ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if ranks > 1:
        import subprocess
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        t0 = time.perf_counter()
        subprocess.check_call([sys.executable, "-m", "wayne_amd.run_visit", "-p", os.path.join(work, "params.yml"),
                               "--gpus", "1", "--ranks-per-gpu", str(ranks)], env=env)
        dt = time.perf_counter() - t0
        out = os.path.join(work, "simulated")
        files = [f for f in os.listdir(out) if f.endswith("_raw.fits")]
        size = sum(os.path.getsize(os.path.join(out, f)) for f in files)
        print("example-shaped visit, %d rank processes on one GPU: %d exposures in %.2f s = %.1f exposures/s end to end "
              "(interpreter start-up of the ranks included), %d FITS files, %.1f MB"
              % (ranks, len(files), dt, len(files) / dt, len(files), size / 1e6))
        return
    t0 = time.perf_counter()
    obs = run_visit.run(["-p", os.path.join(work, "params.yml")])
    dt = time.perf_counter() - t0
    if profile:
        p = list(engine._engines.values())[0].ctx.profile_get()
        n_exp = max(p["k_ramp"]["launches"], 1)
        print("kernels, ms per exposure:", {k: round(v["ms"] / n_exp, 4) for k, v in p.items() if k != "electrons"},
              "electrons per exposure: %.3e" % (p["electrons"] / n_exp))
    files = [f for f in os.listdir(obs.outdir) if f.endswith("_raw.fits")]
    size = sum(os.path.getsize(os.path.join(obs.outdir, f)) for f in files)
    from wayne_amd.exposure_generator import ExposureGenerator
    eg = ExposureGenerator(obs.detector, obs.grism, obs.NSAMP, obs.SAMPSEQ, obs.SUBARRAY)
    K = len(eg._gen_scanning_sample_times(obs.sample_rate)[1])
    print("example-shaped visit: %d exposures (K = %d sub-samples each) in %.2f s = %.1f exposures/s end to end, "
          "%d FITS files, %.1f MB" % (len(files), K, dt, len(files) / dt, len(files), size / 1e6))


if __name__ == "__main__":
    main()
