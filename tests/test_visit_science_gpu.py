"""GPU: visit-level (science-level) parity -- transit depths recovered from whole simulated visits, in ppm.

The reference exists to produce visits from which transit depths are recovered (observation.py:293-357: the
per-wavelength light curves; exposure_generator.py:344-348, 625-628: flux x (1 - depth) -> electrons).  Here a depth
spectrum is injected (0.0146 + 2e-4 sin(2 pi (lambda - 1.1) / 0.3)), visits are generated through the HIP
path in the production mode (split thrower, float32 reads: what `value`, the CLI and the API default run), with float64
reads, per electron with float64 reads (the reference's arithmetic shape) and -- a subset -- in the bit-exact replay
mode, 20 spectral light curves are extracted the way an observer would (tests/visit_science.py) and fitted.

Asserted, every bound a multiple of the fit's own error (from its residuals), none tuned to the measured value:
  (a) recovered - injected within k sigma in every channel, both extractions, every mode -- and the residual scatter is
      the photon noise (x 1.0 ... 1.3 in the median channel): the simulator puts in the light curve it was given and adds no noise of its own;
  (b) production - per-electron, PAIRED (same stellar counts, sky, dark, read-noise draws: the counters do not depend on
      the thrower): every channel within k sigma of zero, the white light curve within k sigma of zero at a sigma of
      under 2 ppm -- the measured replacement of the argued "aggregate approximation budget" of the production thrower;
  (c) float32 against float64 reads: below 0.5 ppm in every channel -- which settles the float32 default;
  (d) replay - per-electron on the subset: the same, at its sigma;
  (e) no dependence of the paired flux ratio on the star's sub-pixel phase in x or y, white light and channel by channel;
  (f) no flux moved between channels (the static part of the pair) -- with a NEGATIVE CONTROL: a library whose
      production throwers drop the fraction of a pixel of every bin's position fails (f) by hundreds of sigma while its
      depths stay right, which is why both are asked for;
  (g) the pairing estimator has unit gain: a visit generated 0.2 % deeper is found 0.2 % deeper.
scripts/visit_science.py runs the same at twice the length and writes profiles/r06/visit_science.json.
"""
import json
import os

import numpy as np
import pytest

import visit_science as vs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "visit_science_test.json")
K_PULL = 4.5          # per channel: 20 channels x 2 extractions x 3 modes x 2 visits = 240 pulls, P(|z| > 4.5) = 7e-6 each
CHI2_20 = 20.0 + 5.0 * np.sqrt(40.0)      # chi2 of 20 channels: mean 20, sigma sqrt(40); 5 sigma = 51.6 (P ~ 1e-4)
CHI2_2 = 18.4         # chi2 of 2 dof at P = 1e-4


def keep(name, rep):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        d = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        d[name] = rep
        json.dump(d, open(REPORT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope="module", params=[("cfg3", 512, 8), ("cfg4", 64, 8)], ids=["cfg3", "cfg4"])
def visit(request):
    name, n, every = request.param
    sv = vs.ScienceVisit(name, n)
    tables = {m: vs.generate(sv, m) for m in ("production", "split_f64", "per_electron")}
    idx = np.arange(0, n, every)
    tables["replay"] = vs.generate(sv, "replay", idx)
    rep = vs.analyse(sv, tables, {"replay": idx})
    keep(name, rep)
    from wayne_amd import engine
    engine.close_all()
    return sv, rep


def test_recovered_depths_are_the_injected_ones(visit):
    sv, rep = visit
    assert 0.40 < sv.G.mean() < 0.65 and sv.G.min() == 0.0 and sv.G.max() == 1.0            # the visit covers the transit
    assert np.ptp(sv.expected) > 3e-4                                                      # ... and the spectrum has its feature
    for mode in ("production", "split_f64", "per_electron"):
        for how in ("ramp", "last_read"):
            r = rep["modes"][mode][how]
            pull = np.array(r["pull"])
            assert np.abs(pull).max() < K_PULL, (mode, how, pull)
            assert r["chi2"] < CHI2_20, (mode, how, r["chi2"])
            assert abs(r["white_recovered_minus_injected_ppm"]) < 4.0 * r["white_sigma_ppm"], (mode, how, r)
            # the scatter about the fitted light curve is the photon noise of the channel (+ sky, read noise, edge pixels)
            # (the end channels sit on the steep flanks of the sensitivity curve, where the two fractional edge columns
            # of a star-fixed channel add their own noise at random sub-pixel phases: the observer's, not the simulator's)
            rms = np.array(r["residual_rms_over_photon_noise"])
            assert 0.85 < rms.min() and np.median(rms) < (1.3 if how == "ramp" else 1.6) and rms.max() < 2.5, (mode, how, rms)


def test_production_mode_against_every_electron_float64_in_ppm(visit):
    sv, rep = visit
    n = rep["n_exposures"]
    for how in ("ramp", "last_read"):
        r = rep["paired"]["production_minus_per_electron"][how]
        d, s = np.array(r["depth_difference_ppm"]), np.array(r["sigma_ppm"])
        assert np.abs(d / s).max() < K_PULL and r["chi2"] < CHI2_20, (how, d, s)
        # what a pair buys: the common noise is gone -- the paired scatter is the partition noise of electrons near a
        # channel's edge, well under the channel's photon noise
        phot = np.array(rep["photon_noise_ppm_per_exposure"])
        assert np.all(np.array(r["paired_flux_rms_ppm"]) < 1.2 * phot)
        # white light: no edges inside, so the pair is nearly noise-free -> the bias bound in ppm
        assert abs(r["white_depth_difference_ppm"]) < 4.0 * r["white_sigma_ppm"], (how, r)
        assert r["white_sigma_ppm"] < 2.0 * np.sqrt(512.0 / n) * (1.0 if sv.v.name == "cfg3" else 0.5), (how, r["white_sigma_ppm"])
        assert abs(r["white_flux_offset_ppm"]) < 4.0 * r["white_flux_offset_sigma_ppm"] + 0.5, (how, r)
        for axis in ("x", "y"):
            ph = r["flux_ratio_vs_%s_phase_ppm" % axis]
            assert ph["chi2"] < CHI2_2, (how, axis, ph)


def test_no_channel_depends_on_the_sub_pixel_phase(visit):
    sv, rep = visit
    # 20 channels x 2 extractions x 2 visits: the largest of 80 chi2 values of 2 dof stays below the 1e-5 quantile
    for how in ("ramp", "last_read"):
        chi2 = np.array(rep["paired"]["production_minus_per_electron"][how]["flux_ratio_vs_x_phase_by_channel_chi2"])
        assert chi2.max() < 23.0, (how, chi2)


def test_no_mode_moves_flux_between_channels(visit):
    # the static part of a pair: the mean flux ratio of every channel over the visit is 1 -- the production thrower puts a
    # channel's electrons where the per-electron thrower puts them (a transit-independent redistribution would cancel in
    # a depth and show here; the negative control below is such a defect)
    sv, rep = visit
    for pair in ("production_minus_per_electron", "replay_minus_per_electron"):
        r = rep["paired"][pair]["ramp"]
        off, sig = np.array(r["channel_flux_offset_ppm"]), np.array(r["channel_flux_offset_sigma_ppm"])
        assert np.abs(off / sig).max() < K_PULL, (pair, off, sig)
        assert np.abs(off).max() < 60.0 * np.sqrt(64.0 / rep["paired"][pair]["n"]) + 15.0, (pair, off)


def test_negative_control_a_dropped_fraction_of_a_pixel_is_seen():
    # The measurement has teeth: a library whose production throwers forget where inside its pixel a bin sits
    # (-DWAYNE_NEGCTL_DROP_FRACTION: the gross form of a position-rounding defect; the replay mode, fp64 positions, is not
    # touched by it) shifts every bin's electrons by its own fraction of a pixel -- half a pixel on average, whatever the
    # star's phase, because 26 bins share a pixel -- so channels on a rising flank of the spectrum gain flux and channels
    # on a falling flank lose it: per-channel flux offsets of thousands of ppm, hundreds of sigma, where the shipped
    # library shows none (above).  A transit DEPTH is blind to it (a static redistribution cancels in the ratio of in- to
    # out-of-transit flux): that is why the pair is asked for both.  In a child process: the library is chosen at load.
    import subprocess
    import sys
    from wayne_amd import build as wb
    lib = wb.build_negctl_fraction()
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r); import visit_science as vs\n"
        "sv = vs.ScienceVisit('cfg3', 96)\n"
        "t = {m: vs.generate(sv, m) for m in ('production', 'replay')}\n"
        "t['per_electron'] = t.pop('replay')\n"           # (pair against the mode the defect does not touch)
        "rep = vs.analyse(sv, t)\n"
        "r = rep['paired']['production_minus_per_electron']['ramp']\n"
        "print(json.dumps({'offset_ppm': r['channel_flux_offset_ppm'], 'sigma_ppm': r['channel_flux_offset_sigma_ppm'],\n"
        "                  'depth_chi2': r['chi2'], 'white_depth_difference_ppm': r['white_depth_difference_ppm'],\n"
        "                  'white_sigma_ppm': r['white_sigma_ppm']}))\n"
    ) % os.path.join(ROOT, "tests")
    env = dict(os.environ, WAYNE_HIP_LIB=lib, WAYNE_ALLOW_FLAGGED_LIB="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    keep("negative_control_dropped_fraction", got)
    off, sig = np.array(got["offset_ppm"]), np.array(got["sigma_ppm"])
    assert np.abs(off).max() > 2000.0 and np.abs(off / sig).max() > 100.0, (off, sig)
    assert off.max() > 500.0 and off.min() < -500.0                      # both signs: flux moved, not lost
    # ... and the depths do not care (which is the point of a differential measurement)
    assert got["depth_chi2"] < CHI2_20 and abs(got["white_depth_difference_ppm"]) < 4.0 * got["white_sigma_ppm"] + 3.0


def test_the_pairing_recovers_an_injected_depth_difference():
    # ... and the estimator has unit gain: the per-electron visit generated with every depth 0.2 % deeper (29 ppm at a
    # depth of 1.46 %), the paired fit must return that difference, white light and channel by channel
    from wayne_amd import engine
    sv = vs.ScienceVisit("cfg3", 256)
    scale = 1.002
    a = vs.generate(sv, "production")
    b = vs.generate(sv, "per_electron", depth_scale=scale)
    engine.close_all()
    for k in (0, 1):
        d, s_, _ = vs.fit_paired(a[k], b[k], sv.G)
        want = -(scale - 1.0) * sv.expected
        assert np.abs((d - want) / s_).max() < K_PULL and (((d - want) / s_) ** 2).sum() < CHI2_20
        dw, sw, _ = vs.fit_paired(vs.white(a[k]), vs.white(b[k]), sv.G)
        want_w = -(scale - 1.0) * sv.white_expected
        assert abs(dw[0] - want_w) < 4.0 * sw[0] and abs(dw[0]) > 10.0 * sw[0]           # seen, and at its size
        keep("injected_difference_%s" % ("ramp" if k == 0 else "last_read"),
             {"injected_ppm": want_w * 1e6, "recovered_ppm": float(dw[0]) * 1e6, "sigma_ppm": float(sw[0]) * 1e6})


def test_float32_reads_change_no_depth(visit):
    sv, rep = visit
    for how in ("ramp", "last_read"):
        a = np.array(rep["paired"]["production_minus_per_electron"][how]["depth_difference_ppm"])
        b = np.array(rep["paired"]["split_f64_minus_per_electron"][how]["depth_difference_ppm"])
        assert np.abs(a - b).max() < 0.5, (how, np.abs(a - b).max())
        wa = rep["paired"]["production_minus_per_electron"][how]["white_depth_difference_ppm"]
        wb = rep["paired"]["split_f64_minus_per_electron"][how]["white_depth_difference_ppm"]
        assert abs(wa - wb) < 0.05, (how, wa, wb)


def test_replay_mode_subset_agrees_too(visit):
    sv, rep = visit
    for how in ("ramp", "last_read"):
        r = rep["paired"]["replay_minus_per_electron"][how]
        d, s = np.array(r["depth_difference_ppm"]), np.array(r["sigma_ppm"])
        assert np.abs(d / s).max() < K_PULL and r["chi2"] < CHI2_20, (how, d, s)
        assert abs(r["white_depth_difference_ppm"]) < 4.0 * r["white_sigma_ppm"], (how, r)
