"""BASELINE.json configs[0] on the reference's OWN inputs: the example visit of ucl-exoplanets/wayne
(examples/hd209458b_12181_simulation_parameters.yml and the five data files it names), committed byte for byte under
tests/fixtures/example_visit/ (data, not source; the stellar-spectrum FITS blob is absent from the reference tree and
both sides use the 6100 K black body the YAML's own comment describes).

CPU: the product's ingestion of the parameter file (run_visit.build_observation; reference run_visit.py:41-320)
against the oracle's independent reading of the same files (oracle/visit_oracle.py, visit_from_parameter_file), and
the numbers SURVEY.md 8 quotes for this visit -- 121 exposures, K = 2233 sub-samples, W = 4494 bins, reads closed by
sub-samples [27, 762, 1497, 2232] -- derived from the FILES.

GPU: `python -m wayne_amd.run_visit -p <that yml> --max-exposures 2` writes the reference's file set; exposures 1-2
through the HIP path against ExposureOracle fed the same files:
  T1  every noise source off, the replay thrower with the YAML's `threads: 4` against the reference's compiled C
      kernel (oracle/_ref) sub-sample by sub-sample: electrons per read interval equal, reads within 1e-4 DN;
  T2  the YAML as it is (stellar Poisson noise, sky, cosmic rays, dark, read noise), production thrower and float32
      reads against the oracle on the same counters.
"""
import hashlib
import os
import shutil

import numpy as np
import pytest
import yaml

from oracle import clib, visit_oracle as vo, wayne_oracle as wo
from wayne_amd import _lib, fitsio, run_visit
from wayne_amd.exposure_generator import ExposureGenerator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "fixtures", "example_visit")
YML = "hd209458b_12181_simulation_parameters.yml"
DATA = ["hd209458b_12181_simulation_" + n for n in ("parameters.yml", "jd.txt", "xref.txt", "yref.txt", "sky.txt",
                                                    "planetary_spectrum.dat")]


def load_cfg():
    with open(os.path.join(EX, YML)) as f:
        return yaml.safe_load(f)


def pair(**noise):
    """The product's Observation and the oracle's twin, both from the fixture files."""
    cfg = load_cfg()
    obs = run_visit.build_observation(cfg, EX)
    if noise:
        obs.setup_noise_sources(**noise)
    det, gr, eo = wo.from_calibration(obs.calibration, "G141", obs.NSAMP, obs.SAMPSEQ, obs.SUBARRAY)
    oo, inp = vo.visit_from_parameter_file(cfg, EX, eo, det)
    return cfg, obs, oo, inp, gr


def test_fixture_is_the_reference_example_byte_for_byte():
    ref = "/root/reference/examples"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    for name in DATA:
        a = hashlib.sha256(open(os.path.join(EX, name), "rb").read()).hexdigest()
        b = hashlib.sha256(open(os.path.join(ref, name), "rb").read()).hexdigest()
        assert a == b, name


def test_visit_numbers_come_out_of_the_files():
    cfg, obs, oo, inp, gr = pair()
    # the five data files
    assert len(obs.exp_start_times) == 121 and obs.x_ref.shape == obs.y_ref.shape == obs.sky_background.shape == (121,)
    for name, got in (("jd", obs.exp_start_times), ("xref", obs.x_ref), ("yref", obs.y_ref), ("sky", obs.sky_background)):
        np.testing.assert_array_equal(got, np.loadtxt(os.path.join(EX, "hd209458b_12181_simulation_%s.txt" % name)))
    assert 403.5 < obs.x_ref.min() < obs.x_ref.max() < 404.6 and 4.7 < obs.sky_background.min() < obs.sky_background.max() < 6.8
    # the YAML
    assert (obs.NSAMP, obs.SAMPSEQ, obs.SUBARRAY, obs.sample_rate, obs.scan_speed, obs.threads) == (5, "SPARS10", 256, 10, 7.4325, 4)
    assert obs.seed == 1963 and obs.cosmic_rate == 11 and obs.x_jitter == 0.025 and obs.y_jitter == 1e-15
    assert type(obs.ssv_gen).__name__ == "SSVSine"
    # the planet spectrum: 15000 rows, sorted, cropped to 0.9-1.8 um by the CLI (run_visit.py:152-153), to the grism's
    # limits by the generator (exposure_generator.py:332) -> W = 4494
    raw = np.loadtxt(os.path.join(EX, "hd209458b_12181_simulation_planetary_spectrum.dat"))
    assert raw.shape == (15000, 2) and raw[0, 0] > raw[-1, 0]                       # (the file runs from 2.0 um down)
    assert obs.wl.shape == (5556,) and np.all(np.diff(obs.wl) > 0) and obs.wl[0] >= 0.9 and obs.wl[-1] <= 1.8
    i0, i1 = wo.crop_spectrum_ind(gr.wl_limits[0], gr.wl_limits[1], obs.wl.copy())
    assert i1 - i0 == 4494
    # sample timing from the mode table and the 10 ms sampling: K = 2233, reads closed by [27, 762, 1497, 2232]
    gen = ExposureGenerator(obs.detector, obs.grism, obs.NSAMP, obs.SAMPSEQ, obs.SUBARRAY, calibration=obs.calibration)
    _, mids, durs, read_index = gen._gen_scanning_sample_times(obs.sample_rate)
    assert len(mids) == len(durs) == 2233 and list(read_index) == [27, 762, 1497, 2232]
    e1 = oo.exposure_inputs(1)
    assert len(e1["sample_mid_points"]) == 2233 and list(e1["read_index"]) == [27, 762, 1497, 2232]
    np.testing.assert_allclose(mids, e1["sample_mid_points"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(durs, e1["sample_durations"], rtol=0, atol=1e-9)
    # product ingestion == the oracle's independent reading of the same files
    np.testing.assert_array_equal(obs.wl, inp["wl"])
    np.testing.assert_array_equal(obs.planet_spectrum, inp["depth"])
    np.testing.assert_allclose(obs.stellar_flux, inp["flux"], rtol=1e-12)
    assert obs.planet.sma_over_rs == pytest.approx(inp["orbit"][1], rel=2e-6)        # (two solar radii: 6e-7 apart)
    np.testing.assert_allclose(obs._visit_trend.scale_factors, oo.scale_factors, rtol=1e-14)
    # (detect_orbits on the file's start times: a short first orbit of 10 exposures, then five of 20-28)
    assert oo.orbit_start_index == list(obs.visit_plan["orbit_start_index"]) == [0, 10, 38, 58, 86, 114]
    # the visit covers the transit: depth ~ 1.46 % mid-visit, ~ 0 at either end
    # (the oracle's light-curve model is a 2-D integration per sample: a few sub-samples of an exposure, not all 2233)
    assert oo.exposure_inputs(1)["planet_signal"].max() < 1e-6
    for number in (1, 61, 121):
        t = oo.exp_start_times[number - 1] + e1["sample_mid_points"][[0, 1100, 2232]] / 86400e3
        want = vo.planet_depths(inp["orbit"], cfg["target"]["ldcoeffs"], inp["depth"], t, inp["rp_white"])
        np.testing.assert_allclose(obs.device_depths(t).host_matrix(), want, rtol=0, atol=2e-8)
        if number == 61:
            assert 0.0140 < want.mean() < 0.0165


@pytest.mark.gpu
def test_cli_on_the_reference_example_writes_its_first_two_exposures(tmp_path):
    work = str(tmp_path / "example")
    shutil.copytree(EX, work)
    obs = run_visit.run(["-p", os.path.join(work, YML), "--max-exposures", "2"])
    assert obs.outdir == os.path.join(work, "hd209458b_12181_data_simulated")
    files = sorted(os.listdir(obs.outdir))
    assert files == ["0000_flt.fits", "0001_raw.fits", "0002_raw.fits", YML, "visit_plan.txt"]
    xs = np.loadtxt(os.path.join(EX, "hd209458b_12181_simulation_xref.txt"))
    jd = np.loadtxt(os.path.join(EX, "hd209458b_12181_simulation_jd.txt"))
    # the same exposures straight from the API (same seed, same exposure index, default options): the files hold them
    cfg, obs2, oo, inp, gr = pair()
    for number in (1, 2):
        h = fitsio.read(os.path.join(obs.outdir, "%04d_raw.fits" % number))
        p0 = h[0].header
        assert len(h) == 1 + 5 * 5 and p0["NSAMP"] == 5 and p0["SAMP_SEQ"] == "SPARS10" and p0["SCAN"] is True
        assert p0["X-REF"] == pytest.approx(xs[number - 1]) and p0["EXPSTART"] == pytest.approx(jd[number - 1] - 2400000.5, abs=1e-6)
        exp = obs2._generate_exposure(obs2.exp_start_times[number - 1], number, write_fits=False)
        reads = [r[0] for r in exp.reads]
        assert reads[0].dtype == np.float32                                          # the default of every entry point
        sci = [hdu.data for hdu in h[1:] if hdu.header.get("EXTNAME") == "SCI"]
        assert len(sci) == 5 and all(s.dtype.kind == "f" and s.dtype.itemsize == 8 for s in sci)
        for k in range(5):                                                           # last read first in the file
            np.testing.assert_array_equal(sci[k], reads[4 - k].astype(np.float64))
        assert 2.5e4 < float(reads[-1].max()) < 4.0e4                                # ~32 000 DN at the trace's brightest pixel
    from wayne_amd import engine
    engine.close_all()


def _device_exposure(obs, number, **options):
    """(reads, record, device depth matrix) of exposure `number` through Observation._generate_exposure."""
    from wayne_amd import engine
    rec = {}
    obs.frame_options = dict(options, record=rec)
    exp = obs._generate_exposure(obs.exp_start_times[number - 1], number, write_fits=False)
    eng = engine.get_engine(obs.device, obs.grism, obs.detector, obs.calibration, obs.NSAMP, obs.SAMPSEQ, obs.SUBARRAY,
                            obs.add_initial_bias)
    depth = eng.ctx.debug_depth(0)
    return np.stack([r[0] for r in exp.reads]), rec, depth


@pytest.mark.gpu
@pytest.mark.parametrize("number", [1, 2, 61])
def test_example_exposures_noise_off_against_the_reference_c(number):
    # (exposures 1 and 2 are what BASELINE configs[0] names; 61 is the middle of the transit, depth 1.46 %)
    if not clib.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference when the checker was built)")
    off = dict(sky_background=0.0, cosmic_rate=None, add_read_noise=False, add_stellar_noise=False)
    cfg, obs, oo, inp, gr = pair(**off)
    obs.add_dark = False
    threads = cfg["general"]["threads"]
    assert threads == 4
    got, rec, depth_dev = _device_exposure(obs, number, rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64, threads=threads)
    # the device's light curves against the oracle's own model (an independent 2-D integration, oracle/lc_oracle.c): the
    # whole matrix out of transit, three of the 2233 sub-samples in it (the 2-D integration takes minutes for all of them)
    W0 = wo.crop_spectrum_ind(gr.wl_limits[0], gr.wl_limits[1], inp["wl"].copy())
    want_inp = oo.exposure_inputs(number, with_depths=number != 61)
    if number != 61:
        np.testing.assert_allclose(depth_dev, want_inp["planet_signal"][:, W0[0]:W0[1]], rtol=0, atol=3e-8)
        signal = want_inp["planet_signal"].copy()
    else:
        rows = [0, 1100, 2232]
        own = vo.planet_depths(inp["orbit"], cfg["target"]["ldcoeffs"], inp["depth"], want_inp["time_array"][rows], inp["rp_white"])
        np.testing.assert_allclose(depth_dev[rows], own[:, W0[0]:W0[1]], rtol=0, atol=1.5e-7)
        assert 0.0140 < depth_dev.mean() < 0.0165
        signal = np.zeros((2233, inp["wl"].size))
    signal[:, W0[0]:W0[1]] = depth_dev             # ... then handed over, so that np.round sees the same means to the last bit
    orec = {}
    want = np.stack(oo.generate_exposure(number, wo.PhiloxDraws(obs.seed, number - 1, 256), thrower="ref", record=orec,
                                         planet_signal=signal, add_dark=False, **off))
    counts_o, acc_o = np.stack(orec["counts"]), np.stack(orec["acc"])
    assert counts_o.shape == (2233, 4494) and 1.9e7 < counts_o.sum() < 2.4e7 and counts_o.max() < 32
    np.testing.assert_array_equal(rec["counts"], counts_o)
    np.testing.assert_allclose(rec["x"], np.stack(orec["x"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(rec["y"], np.stack(orec["y"]), rtol=0, atol=1e-9)
    # per-exposure host draws: the jittered star positions and the reference's `test` seeds of the 2233 thrower calls
    assert rec["seeds"].shape == (2233,) and rec["seeds"].min() >= 0 and rec["seeds"].max() < 100000
    # electrons per read interval: the reference's C kernel, called 2233 times with threads = 4, and the device's
    # replay thrower put every electron on the same pixel (flat-weighted sums: 2^-28 e- per tile flush)
    np.testing.assert_allclose(rec["acc"], acc_o, rtol=1e-13, atol=4096.0 * 2233 * 2.0 ** -29)
    assert float(np.abs(rec["acc"] - acc_o).max()) < 1e-3
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-4)
    assert got.shape == (5, 266, 266) and 2.5e4 < got[-1].max() < 4.0e4
    from wayne_amd import engine
    engine.close_all()


@pytest.mark.gpu
@pytest.mark.parametrize("number", [1, 2])
def test_example_exposures_as_the_yaml_has_them_on_the_same_counters(number):
    # every switch of the YAML on; production thrower (thin bins: every electron from its bin's own lane), production
    # math, float32 reads -- what the CLI wrote above -- against the oracle driven by the same Philox counters
    import test_fullsize_oracle_gpu as fs
    cfg, obs, oo, inp, gr = pair()
    got, rec, depth_dev = _device_exposure(obs, number)
    assert got.dtype == np.float32
    want_inp = oo.exposure_inputs(number)
    W0 = wo.crop_spectrum_ind(gr.wl_limits[0], gr.wl_limits[1], inp["wl"].copy())
    signal = want_inp["planet_signal"].copy()
    signal[:, W0[0]:W0[1]] = depth_dev
    orec = {}
    draws = wo.PhiloxDraws(obs.seed, number - 1, 256)
    want = np.stack(oo.generate_exposure(number, draws, thrower="split", record=orec, planet_signal=signal))
    orec = {k: np.stack(orec[k]) for k in ("counts", "acc")}
    dts = np.diff(np.concatenate([[0.0], oo.eo.read_times]))
    cosmic = wo.PhiloxDraws(obs.seed, number - 1, 256)
    for r in range(4):                       # (the device's accumulators hold the cosmic-ray hits: see fs.oracle_exposure)
        orec["acc"][r, 5:-5, 5:-5] += cosmic.cosmic_frame(cfg["observation"]["cosmic_rate"], dts[r], 256, r)
    fs.compare("example_visit_%d" % number, "production_f32", got, rec, want, orec, 3e-5, 3e-3, 1e-4, med_rel=1.2e-7)
    from wayne_amd import engine
    engine.close_all()
