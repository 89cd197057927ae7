"""Real-file readiness without the files: the CALWF3 / aXe calibration files the reference downloads at import
(params.py:20-56) are not in this container, so a directory in the layout the reference's CODE opens is written by an
independent encoder (tests/stsci_files.py: its own FITS writer, STScI's conventions where the reference depends on
them, everything else deliberately varied) and read by the product (CalibrationSet.from_directory; the CLI's
--calibration DIR).

CPU: every plane, table and header value arrives bit for bit; which HDU belongs to which read of a super-dark.
GPU: the mini visit through `run_visit --calibration DIR` against ExposureOracle built from the SOURCE arrays (not from
the product's reading of the files): deterministic tier to 1e-4 DN, noisy tier on the same counters.
Reference: grism.py:66-76, 79-80, 97-106, 411-423; detector.py:31, 56-67, 183-190, 200-209.
"""
import os
import shutil

import numpy as np
import pytest
import yaml

import stsci_files
from oracle import wayne_oracle as wo
from wayne_amd import _lib, calibration, detector, fitsio, run_visit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINI = os.path.join(ROOT, "tests", "fixtures", "mini_visit")
MODES = [(128, "RAPID"), (256, "SPARS10")]


@pytest.fixture(scope="module")
def cal_dir(tmp_path_factory):
    src = calibration.CalibrationSet.synthetic(23)
    det = detector.WFC3_IR()
    d = str(tmp_path_factory.mktemp("calb"))
    stsci_files.write_calibration_directory(d, src, det, MODES, grisms=("G141", "G102"))
    return src, det, d


def test_files_are_not_the_products_writer(cal_dir):
    # the point of the exercise: these bytes differ from what wayne_amd.fitsio would write for the same content
    src, det, d = cal_dir
    raw = open(os.path.join(d, calibration.FLAT_FILES["G141"]), "rb").read()
    assert len(raw) % 2880 == 0 and b"WMIN    =" in raw[:2880 * 2] and b"D+04" in raw[:2880 * 2]      # a D exponent
    h = fitsio.read(os.path.join(d, calibration.FLAT_FILES["G141"]))
    assert len(h) == 5 and h[0].data.dtype.itemsize == 4                # a fifth HDU the reference never opens
    own = str(os.path.join(d, "own.fits"))
    fitsio.write(own, [fitsio.HDU(fitsio.Header([("WMIN", 10600.0, ""), ("WMAX", 17000.0, "")]), src.flat["G141"][0])])
    assert open(own, "rb").read()[:2880] != raw[:2880]
    os.remove(own)


def test_from_directory_reads_the_references_layout(cal_dir):
    src, det, d = cal_dir
    assert sorted(os.listdir(d)) == sorted(
        [calibration.FLAT_FILES[g] for g in ("G141", "G102")] + [calibration.SKY_FILES[g] for g in ("G141", "G102")] +
        [calibration.SENS_FILES[g] for g in ("G141", "G102")] + [calibration.PFL_FILE, calibration.LIN_FILE] +
        [det.dark_file(*m) for m in MODES])
    cal = calibration.CalibrationSet.from_directory(d, det)
    for g in ("G141", "G102"):
        assert cal.flat[g].dtype == np.float32 and np.array_equal(cal.flat[g], src.flat[g])          # HDUs 0..3 (grism.py:73-76)
        assert cal.flat_wl[g] == src.flat_wl[g]                                                      # primary header (:71-72)
        assert np.array_equal(cal.sky[g], src.sky[g])                                                # HDU 0 (:417-418)
        np.testing.assert_allclose(cal.sens[g][0], src.sens[g][0], rtol=0, atol=1e-12)               # angstrom -> micron (:102-103)
        assert np.array_equal(cal.sens[g][1], src.sens[g][1])
    assert cal.pfl.shape == (1014, 1014) and np.array_equal(cal.pfl, src.pfl)                        # HDU 1 [5:-5, 5:-5] (detector.py:203)
    assert np.array_equal(cal.lin, src.lin)                                                          # HDUs 1..4 (detector.py:58-67)
    for subarray, sampseq in MODES:
        for NSAMP in (2, 4, 9, 16):
            rt = det.get_read_times(NSAMP, subarray, sampseq)
            got, want = cal.dark_frames(subarray, sampseq, rt, det), src.dark_frames(subarray, sampseq, rt, det)
            assert got[0].shape == (NSAMP - 1, subarray + 10, subarray + 10)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # read with NSAMP index n: HDU -5 n, its error the next one (detector.py:185-190), straight off the file
        h = fitsio.read(os.path.join(d, det.dark_file(subarray, sampseq)))
        assert len(h) == 1 + 5 * 16 and str(h[-5 * 3].header["EXTNAME"]).strip() == "SCI"
        rt = det.get_read_times(4, subarray, sampseq)
        assert np.array_equal(h[-5 * 3].data, cal.dark_frames(subarray, sampseq, rt, det)[0][1])
        assert np.array_equal(h[-5 * 3 + 1].data, cal.dark_frames(subarray, sampseq, rt, det)[1][1])
    # the planes the context is handed for a mode are those of the source set
    rt = det.get_read_times(4, 128, "RAPID")
    a, b = cal.for_mode("G141", 128, "RAPID", rt, detector=det), src.for_mode("G141", 128, "RAPID", rt, detector=det)
    assert sorted(a) == sorted(b)
    for k in a:
        if isinstance(a[k], list):
            assert all(np.array_equal(x, y) for x, y in zip(a[k], b[k])), k
        elif isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
def test_cli_with_a_calibration_directory_against_the_oracle(cal_dir, tmp_path):
    src, det, d = cal_dir
    work = str(tmp_path / "visit")
    shutil.copytree(MINI, work)
    yml = os.path.join(work, "params.yml")
    obs = run_visit.run(["-p", yml, "--calibration", d, "--max-exposures", "2", "--float64-reads"])
    assert obs.calibration is not src and not getattr(obs.calibration, "_synthetic_dark", True)
    files = [os.path.join(obs.outdir, "%04d_raw.fits" % n) for n in (1, 2)]
    # the oracle over the SOURCE arrays: nothing of from_directory on its side
    cfg = yaml.safe_load(open(yml))
    import test_example_visit as tev
    from oracle import visit_oracle as vo
    _, gr, eo = wo.from_calibration(src, "G141", obs.NSAMP, obs.SAMPSEQ, obs.SUBARRAY)
    # (the mini visit's stellar spectrum is a black body of `star_temperature`; the example visit's reader covers the keys)
    oo, inp = vo.visit_from_parameter_file(cfg, work, eo, det, star_temperature=cfg["target"]["star_temperature"])
    N = obs.SUBARRAY
    from wayne_amd import engine
    for number in (1, 2):
        # deterministic tier + same-counter noise, replay thrower: through the API of the SAME observation object
        off = dict(cosmic_rate=None, add_read_noise=False, add_stellar_noise=False)
        obs.setup_noise_sources(obs.sky_background, **off)
        got, rec, depth_dev = tev._device_exposure(obs, number, rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                   exact_samplers=True, threads=2)
        want_inp = oo.exposure_inputs(number)
        W0 = wo.crop_spectrum_ind(gr.wl_limits[0], gr.wl_limits[1], inp["wl"].copy())
        signal = want_inp["planet_signal"].copy()
        # (exposure 1 of the mini visit is in ingress, where the depth moves by 4e-8 for the 6e-7 by which the oracle's
        # a / R* -- its own solar radius -- differs from the product's; the device's model itself is good to 2e-8)
        np.testing.assert_allclose(depth_dev, signal[:, W0[0]:W0[1]], rtol=0, atol=1.5e-7)
        signal[:, W0[0]:W0[1]] = depth_dev
        want = np.stack(oo.generate_exposure(number, wo.PhiloxDraws(obs.seed, number - 1, N), thrower="oracle",
                                             planet_signal=signal, threads=2, **off))
        d_ = np.abs(got - want)
        bad = int((d_ > 1e-3 + 1e-6 * np.abs(want)).sum())
        assert bad <= 2e-3 * got.size and np.median(d_) < 1e-4 and got[-1].max() > 10, (number, bad)
        # ... and the file the CLI wrote holds the production exposure of this calibration set: same visit, default
        # switches, float64 reads -- regenerated here through the API, bit for bit
        obs.setup_noise_sources(obs.sky_background, cfg["observation"]["cosmic_rate"], cfg["observation"]["add_read_noise"],
                                cfg["observation"]["add_stellar_noise"])
        obs.frame_options = {"out_dtype": np.float64}
        exp = obs._generate_exposure(obs.exp_start_times[number - 1], number, write_fits=False)
        sci = [h.data for h in fitsio.read(files[number - 1])[1:] if str(h.header.get("EXTNAME", "")).strip() == "SCI"]
        for k, (r, _) in enumerate(exp.reads):
            np.testing.assert_array_equal(sci[len(sci) - 1 - k], r)
    engine.close_all()
