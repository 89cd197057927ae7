"""The visit driver: FITS reader / writer, visit planner,
Observation + YAML front end.  CPU tests cover the host logic; the GPU test runs the
CLI on a small visit and checks the files against direct ExposureGenerator calls."""
import os

import numpy as np
import pytest
import yaml

from wayne_amd import detector, fitsio, observation, run_visit, tools, visit_planner

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINI = os.path.join(HERE, "fixtures", "mini_visit")


def test_fits_roundtrip_all_types(tmp_path):
    rng = np.random.RandomState(0)
    arrays = [rng.normal(size=(7, 5)), rng.normal(size=(3, 4)).astype(np.float32),
              rng.randint(-1000, 1000, (4, 6)).astype(np.int16), rng.randint(0, 255, (2, 3)).astype(np.uint8),
              rng.randint(-10 ** 6, 10 ** 6, (5, 2)).astype(np.int32)]
    hdus = [fitsio.HDU(fitsio.Header([("WMIN", 10600.0, "angstrom"), ("NAME", "it's", ""), ("FLAG", True, "")]), None)]
    hdus += [fitsio.HDU(fitsio.Header([("SAMPNUM", i, "")]), a, name="SCI") for i, a in enumerate(arrays)]
    p = str(tmp_path / "t.fits")
    fitsio.write(p, hdus)
    assert os.path.getsize(p) % 2880 == 0
    back = fitsio.read(p)
    assert back[0].header["WMIN"] == 10600.0 and back[0].header["NAME"] == "it's" and back[0].header["FLAG"] is True
    for i, a in enumerate(arrays):
        assert back[i + 1].name == "SCI" and back[i + 1].header["SAMPNUM"] == i
        np.testing.assert_array_equal(back[i + 1].data, a)
        assert back[i + 1].data.dtype.kind == a.dtype.kind and back[i + 1].data.dtype.itemsize == a.dtype.itemsize


def test_exposure_file_layout_and_primary_header(tmp_path):
    # the reference's file: primary header, then per read (latest first) SCI + four empty extensions
    # (exposure.py:133-214), primary keywords as exposure.py:216-410 writes them
    from wayne_amd import calibration, exposure, grism
    from wayne_amd.exposure_generator import ExposureGenerator
    from wayne_amd.observation import Planet
    cal = calibration.CalibrationSet.synthetic(11)
    det, gr = detector.WFC3_IR(), grism.G141(cal)
    pl = Planet("HD 209458 b", period=3.524746, sma_au=0.047309, stellar_radius_rsun=1.155, inclination=86.71,
                transittime=2456196.28836)
    eg = ExposureGenerator(det, gr, 4, "RAPID", 128, pl, "0007_raw.fits", 2456196.25, calibration=cal, seed=77)
    info = dict(eg.exp_info, x_ref=401.5, y_ref=399.0, samp_rate=25.0, SCAN=True, SCAN_DIR=1, sky_background=1.3,
                cosmic_rate=11.0, scale_factor=0.9991, add_dark=True, sim_time=0.002)
    exp = exposure.Exposure(det, gr, pl, info)
    rng = np.random.default_rng(0)
    frames = [rng.normal(100 * r, 5, (138, 138)).astype(np.float32) for r in range(4)]
    t = det.get_read_times(4, 128, "RAPID")
    exp.add_read(frames[0], {"cumulative_exp_time": 0.0, "read_exp_time": 0.0, "CRPIX1": 0})
    for r in range(3):
        exp.add_read(frames[r + 1], {"cumulative_exp_time": float(t[r]), "read_exp_time": float(t[r] - (t[r - 1] if r else 0)),
                                     "CRPIX1": 0})
    path = exp.generate_fits(str(tmp_path), ldcoeffs=[0.8, -0.7, 0.9, -0.4])
    assert os.path.basename(path) == "0007_raw.fits"
    h = fitsio.read(path)
    assert [x.name for x in h[1:6]] == ["SCI", "ERR", "DQ", "SAMP", "TIME"] and len(h) == 1 + 5 * 4
    sci = [x for x in h if x.name == "SCI"]
    assert [x.header["SAMPNUM"] for x in sci] == [3, 2, 1, 0]
    assert sci[0].data.dtype == np.dtype(">f8") or sci[0].data.dtype == np.float64
    np.testing.assert_array_equal(sci[0].data, frames[3].astype(np.float64))
    np.testing.assert_array_equal(sci[3].data, frames[0].astype(np.float64))
    assert sci[0].header["SAMPTIME"] == pytest.approx(t[2]) and sci[3].header["SAMPTIME"] == 0.0
    # the read header's three cards (exposure.py:413-430) and SAMPNUM (:161) are the reference's; EXTVER (1 for the last
    # read, counting up, on all five extensions of a read) and BUNIT = COUNTS are this writer's additions, as in a real
    # _raw file -- a superset a reader of the reference's files never misses
    assert sci[0].header["DELTATIM"] == pytest.approx(t[2] - t[1]) and sci[2].header["DELTATIM"] == pytest.approx(t[0])
    assert [x.header["CRPIX1"] for x in sci] == [0, 0, 0, 0]
    assert [x.header["EXTVER"] for x in sci] == [1, 2, 3, 4] and all(x.header["BUNIT"] == "COUNTS" for x in sci)
    assert [x.header["EXTVER"] for x in h[1:6]] == [1] * 5 and [x.header["EXTVER"] for x in h[16:21]] == [4] * 5
    p0 = h[0].header
    assert p0["TELESCOP"] == "HST" and p0["INSTRUME"] == "WFC3" and p0["DETECTOR"] == "IR" and p0["FILTER"] == "G141"
    assert p0["EXPSTART"] == pytest.approx(2456196.25 - 2400000.5, abs=1e-9) and p0["EXPTIME"] == pytest.approx(t[-1])
    assert p0["SUBARRAY"] is True and p0["SUBTYPE"] == "SQ128SUB" and p0["APERTURE"] == "GRISM128" and p0["NSAMP"] == 4
    assert p0["SAMP_SEQ"] == "RAPID" and p0["OBSMODE"] == "MULTIACCUM" and p0["POSTARG2"] == 1
    assert p0["X-REF"] == 401.5 and p0["STARX"] == 401.5 and p0["SAMPRATE"] == pytest.approx(0.025)
    assert p0["ADD-DRK"] is True and p0["ADD-FLAT"] is False and p0["CSMCRATE"] == 11.0 and p0["SKY-LVL"] == 1.3
    assert p0["VSTTREND"] == pytest.approx(0.9991) and p0["RANDSEED"] == 77 and p0["SIM"] is True
    assert p0["TARGNAME"] == "HD 209458 b" and p0["PERIOD"] == pytest.approx(3.524746) and p0["INC"] == pytest.approx(86.71)
    assert p0["SMA"] == pytest.approx(0.047309 / (1.155 * 0.00465047), rel=1e-3) and p0["LD2"] == pytest.approx(-0.7)


def test_fits_reads_reference_data_files():
    # the reference's own small FITS data files, when its tree is present (not on the GPU box)
    ref = "/root/reference/wayne/data"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    bias = fitsio.read(os.path.join(ref, "wfc3_ir_initial_bias_256.fits"))[1].data
    np.testing.assert_array_equal(bias, np.load(detector.WFC3_IR().initial_bias))
    tab = fitsio.read(os.path.join(ref, "wfc3_ir_g141_src_004_syn.fits"))[1].data
    assert tab.dtype.names[:2] == ("WAVELENGTH", "THROUGHPUT") and len(tab) == 87
    assert 9000 < tab["WAVELENGTH"].min() < tab["WAVELENGTH"].max() < 20000 and np.all(np.diff(tab["WAVELENGTH"]) > 0)


def test_calibration_from_directory_roundtrip(tmp_path):
    from wayne_amd import calibration
    syn = calibration.CalibrationSet.synthetic(5, grisms=("G141",))
    d = str(tmp_path)
    H, hdr = fitsio.HDU, fitsio.Header
    fitsio.write(os.path.join(d, calibration.FLAT_FILES["G141"]),
                 [H(hdr([("WMIN", 10600.0, ""), ("WMAX", 17000.0, "")]), syn.flat["G141"][0])] +
                 [H(hdr(), syn.flat["G141"][i]) for i in (1, 2, 3)])
    fitsio.write(os.path.join(d, calibration.SKY_FILES["G141"]), [H(hdr(), syn.sky["G141"])])
    pfl = np.ones((1024, 1024), dtype=np.float32)
    pfl[5:-5, 5:-5] = syn.pfl
    fitsio.write(os.path.join(d, calibration.PFL_FILE), [H(hdr(), None), H(hdr(), pfl)])
    fitsio.write(os.path.join(d, calibration.LIN_FILE), [H(hdr(), None)] + [H(hdr(), syn.lin[i]) for i in range(4)])
    got = calibration.CalibrationSet.from_directory(d)
    np.testing.assert_array_equal(got.flat["G141"], syn.flat["G141"])
    assert got.flat_wl["G141"] == (10600.0, 17000.0)
    np.testing.assert_array_equal(got.sky["G141"], syn.sky["G141"])
    np.testing.assert_array_equal(got.pfl, syn.pfl)
    np.testing.assert_array_equal(got.lin, syn.lin)


def test_detect_orbits_and_helpers():
    assert tools.detect_orbits([1.001, 1.002, 1.032]) == [0, 2]            # tests/test_tools.py:85-89
    g = tools.wl_at_resolution(130, 1.0, 1.7)
    assert abs(np.diff(g)[0] - 1.35 / 130) < 1e-12 and g[0] == 1.0 and g[-1] >= 1.7
    wl = np.linspace(1, 2, 2001)
    new = np.linspace(1.1, 1.9, 17)
    np.testing.assert_allclose(tools.rebin_spec(wl, 3 + 2 * wl, new), 3 + 2 * new, rtol=1e-9)   # linear: exact
    flat = tools.rebin_spec(wl, np.full(wl.size, 7.0), new)
    np.testing.assert_allclose(flat, 7.0)
    bb = tools.blackbody_lambda(np.array([0.5, 1.0, 2.0]), 6100.0)
    assert bb[0] > bb[1] > bb[2] > 0                                        # Wien peak at 0.475 um


def test_rebin_spec_against_the_oracle():
    # the product's cumulative-integral rebin against the oracle's bin-by-bin statement of pysynphot's published
    # binning (tools.py:131-149): irregular input grid, output bins finer and coarser than it, bins that run off
    # the sampled range, a spectrum with lines; and flux conservation over the common range
    from oracle import wayne_oracle as wo
    rng = np.random.default_rng(12)
    wl = np.sort(rng.uniform(0.9, 1.8, 1500))
    sp = 5 + np.sin(30 * wl) + 3 * np.exp(-((wl - 1.3) / 0.002) ** 2) + rng.normal(0, 0.1, wl.size)
    for new in (np.linspace(0.95, 1.75, 4000), np.linspace(1.0, 1.7, 37), np.sort(1 / (0.55 + 1e-3 * np.arange(500))),
                np.linspace(0.85, 1.85, 60)):
        got, want = tools.rebin_spec(wl, sp, new), wo.rebin_spec(wl, sp, new)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)
    new = np.linspace(1.0, 1.7, 200)
    e = tools.bin_centers_to_edges(new)
    inside = (wl >= e[0]) & (wl <= e[-1])
    grid = np.concatenate([[e[0]], wl[inside], [e[-1]]])
    total = np.trapezoid(np.interp(grid, wl, sp), grid)
    assert abs((tools.rebin_spec(wl, sp, new) * np.diff(e)).sum() - total) < 1e-10 * total


def _bintable_fits(path, columns):
    """A minimal FITS file with an empty primary HDU and one BINTABLE extension, written byte by byte (FITS 4.0
    section 7.3): columns = [(name, TFORM letter, values)] -- the layout of the PHOENIX grid files the reference
    reads with astropy (tools.py:152-170)."""
    def card(k, v, quote=False):
        val = ("'%-8s'" % v) if quote else ("%20s" % (("T" if v else "F") if isinstance(v, bool) else v))
        return ("%-8s= %s" % (k, val)).ljust(80).encode("ascii")

    def block(cards):
        raw = b"".join(cards) + "END".ljust(80).encode("ascii")
        return raw + b" " * (-len(raw) % 2880)

    sizes = {"D": (">f8", 8), "E": (">f4", 4), "J": (">i4", 4)}
    nrows = len(columns[0][2])
    rowlen = sum(sizes[f][1] for _, f, _ in columns)
    rows = np.zeros(nrows, dtype=np.dtype([(n, sizes[f][0]) for n, f, _ in columns]))
    for n, f, v in columns:
        rows[n] = v
    primary = block([card("SIMPLE", True), card("BITPIX", 8), card("NAXIS", 0), card("EXTEND", True)])
    cards = [card("XTENSION", "BINTABLE", True), card("BITPIX", 8), card("NAXIS", 2), card("NAXIS1", rowlen),
             card("NAXIS2", nrows), card("PCOUNT", 0), card("GCOUNT", 1), card("TFIELDS", len(columns))]
    for i, (n, f, _) in enumerate(columns, 1):
        cards += [card("TTYPE%d" % i, n, True), card("TFORM%d" % i, f, True)]
    data = rows.tobytes()
    with open(path, "wb") as fh:
        fh.write(primary + block(cards) + data + b"\0" * (-len(data) % 2880))


def test_phoenix_grid_loader_sorts_and_drops_duplicates(tmp_path):
    # tools.py:152-170: order by wavelength, then keep idx = nonzero(diff(wl)) -- which drops the first of every run
    # of equal wavelengths AND, as written in the reference, the last sample of the file
    wl = np.array([1.30, 1.10, 1.20, 1.20, 1.70, 1.00, 1.50, 1.50, 1.50, 1.60])
    fl = np.arange(10, dtype=float) * 1e5 + 3.0
    p = str(tmp_path / "lte.fits")
    _bintable_fits(p, [("Wavelength", "D", wl), ("Flux", "D", fl), ("Extra", "J", np.arange(10))])
    got_wl, got_fl = tools.load_pheonix_stellar_grid_fits(p)
    order = np.argsort(wl, kind="stable")
    swl, sfl = wl[order], fl[order]
    keep = np.nonzero(np.diff(swl))
    np.testing.assert_array_equal(got_wl, swl[keep])
    np.testing.assert_array_equal(got_fl, sfl[keep])
    assert np.all(np.diff(got_wl) > 0) and got_wl[-1] == 1.60 and got_wl.size == 6
    # float32 columns and other capitalisation of the names load the same way
    _bintable_fits(p, [("WAVELENGTH", "E", wl), ("FLUX", "E", fl)])
    w32, f32 = tools.load_pheonix_stellar_grid_fits(p)
    np.testing.assert_allclose(w32, got_wl, rtol=1e-7)
    np.testing.assert_allclose(f32, got_fl, rtol=1e-7)


def test_visit_planner():
    det = detector.WFC3_IR()
    vp = visit_planner.VisitPlanner(det, 5, "SPARS10", 256, num_orbits=3)
    t = vp["exp_times"]
    assert vp["num_exp"] == len(t) and vp["orbit_start_index"][0] == 0 and len(vp["orbit_start_index"]) == 3
    assert t[0] == 6.0 and t[vp["orbit_start_index"][1]] == 95.0 + 5.0      # guide-star acquisition 6 / 5 min
    step = det.exptime(5, 256, "SPARS10") / 60.0 + 1.0
    assert abs(t[1] - t[0] - step) < 1e-12
    assert np.all(t[:vp["orbit_start_index"][1]] < 54.0)                    # visibility window
    assert det.num_exp_per_buffer(5, 256) == 21                              # floor(2*16*4 / 6)
    # buffer dumps (visit_planner.py:74-76, 106-111): `exp_n > exp_per_dump` -- after exp_per_dump + 1 exposures -- the visit
    # waits 5.8 minutes and the count starts again.  A full-array NSAMP 16 exposure fills the buffer by itself: every second
    # step is a step + 5.8 min (the mutation audit's 8.5-minute dump was stopped only by the oracle's planner)
    assert det.num_exp_per_buffer(16, 1024) == 1
    vp = visit_planner.VisitPlanner(det, 16, "SPARS10", 1024, num_orbits=2)
    t, first = vp["exp_times"], vp["orbit_start_index"][1]
    step = det.exptime(16, 1024, "SPARS10") / 60.0 + 1.0
    gaps = np.diff(t[:first])
    np.testing.assert_allclose(gaps[0::2], step, rtol=0, atol=1e-12)
    np.testing.assert_allclose(gaps[1::2], step + 5.8, rtol=0, atol=1e-12)
    assert vp["buffer_dump_index"][:3] == [2, 4, 6]
    assert t[first] == 95.0 + 5.0 and abs(t[first + 1] - t[first] - step) < 1e-12            # the count restarts with every orbit (:97)


def test_build_observation_from_yaml():
    cfg = yaml.safe_load(open(os.path.join(MINI, "params.yml")))
    obs = run_visit.build_observation(cfg, MINI)
    assert len(obs.exp_start_times) == 6 and obs.visit_plan["orbit_start_index"] == [0, 2, 5]
    assert obs.NSAMP == 4 and obs.SUBARRAY == 128 and obs.grism.name == "G141"
    assert obs.transmission_spectroscopy and obs.ssv_gen.stddev == 1.5
    assert obs.wl.min() >= 0.9 and obs.wl.max() <= 1.8                       # run_visit.py:152-153 pre-crop
    raw_wl = np.sort(np.loadtxt(os.path.join(MINI, "planet_spectrum.dat"))[:, 0])
    assert obs.wl.min() == raw_wl[raw_wl >= 0.9].min() and obs.wl.max() == raw_wl[raw_wl <= 1.8].max()    # ... and nothing narrower
    # no stellar file: a black body of the star's temperature on the planet's grid, times flux_scale ONCE (run_visit.py:201-205)
    from wayne_amd import tools
    np.testing.assert_allclose(obs.stellar_flux, tools.blackbody_lambda(obs.wl, 6100.0) * 1.8e-20, rtol=1e-12)
    assert obs.sample_rate == 25 and obs.scan_speed == 30.0                  # ms, px/s, as the YAML gives them
    assert obs._visit_trend.scale_factors.shape == (6,)
    t, model = obs.show_lightcurve()
    assert model[0] > 0.995 and model[3] < 0.986 and model[5] > 0.99         # ingress, mid-transit, out
    dd = obs.device_depths(obs.exp_start_times[3] + np.array([0.0, 1e-4]))
    m = dd.host_matrix()
    assert m.shape == (2, obs.wl.size) and 0.015 < m.mean() < 0.0175


@pytest.mark.gpu
def test_generate_exposure_follows_the_references_bookkeeping():
    # Observation._generate_exposure and setup_visit as the reference writes them (observation.py:189-226, 415-462), host half
    # only (ExposureGenerator.prepare: nothing is launched; the context exists for the calibration it holds) -- the mutation audit's four visit-level mutants (sample times as seconds,
    # one shift too many, the visit trend ignored, planner minutes / 86400) passed everything but the visit ORACLE:
    #   exp_start_times = planner minutes -> days + start_JD;   file number n -> index n - 1;
    #   x_ref / y_ref = their per-exposure value + shift x index;   sky and visit-trend factor of that index;
    #   the light curves asked for at expstart + sub-sample mid-points (ms -> days)
    from wayne_amd import observation, visit_planner
    cfg = yaml.safe_load(open(os.path.join(MINI, "params.yml")))
    obs = run_visit.build_observation(cfg, MINI)
    obs.x_shifts, obs.y_shifts = 0.37, -0.11
    asked = []
    inner = obs.device_depths

    def recording(time_array):
        asked.append(np.array(time_array, dtype=float))
        return inner(time_array)

    obs.device_depths = recording
    xs_, ys_, sky = (np.loadtxt(os.path.join(MINI, f)) for f in ("xref.txt", "yref.txt", "sky.txt"))
    jd = np.loadtxt(os.path.join(MINI, "jd.txt"))
    np.testing.assert_array_equal(obs.exp_start_times, jd)                       # (times given: taken as they are, :202-204)
    t = np.asarray(obs.visit_plan["exp_start_times"], dtype=float)
    t0 = np.concatenate([np.full(hi - lo, t[lo]) for lo, hi in zip(obs.visit_plan["orbit_start_index"],
                                                                   obs.visit_plan["orbit_start_index"][1:] + [len(t)])])
    a1, b1, b2, to = cfg["trends"]["visit_trend_coeffs"]
    trend = (1 - a1 * (t - to)) * (1 - b1 * np.exp(-b2 * (t - t0)))             # visit_trends.py:44-57
    for n in (1, 2, 4, 6):
        asked.clear()
        g = obs._generate_exposure(obs.exp_start_times[n - 1], n, prepare_only=True)
        i = n - 1
        assert g.exp_info["filename"] == "%04d_raw.fits" % n
        assert g.exp_info["x_ref"] == pytest.approx(np.atleast_1d(xs_)[i] + 0.37 * i, abs=1e-12)
        assert g.exp_info["y_ref"] == pytest.approx(np.atleast_1d(ys_)[i] - 0.11 * i, abs=1e-12)
        assert g.exp_info["sky_background"] == pytest.approx(np.atleast_1d(sky)[i])
        assert g.exp_info["scale_factor"] == pytest.approx(trend[i], rel=1e-12) and abs(trend[i] - 1.0) > 1e-5
        _, mid, _, _ = g._gen_scanning_sample_times(obs.sample_rate)
        assert len(asked) == 1
        np.testing.assert_allclose(asked[0], jd[i] + np.asarray(mid) / 86400e3, rtol=0, atol=1e-12)
        assert 1e-7 < asked[0][-1] - asked[0][0] < 1e-3                          # (a fraction of a second to a minute, in days)
    # a generated plan: the planner's minutes (exposure overhead 3 min, :215-219) over 1440, plus start_JD
    o2 = observation.Observation()
    o2.setup_detector(obs.detector, 4, "RAPID", 128)
    o2.setup_visit(2456196.25, 2)
    vp = visit_planner.VisitPlanner(obs.detector, 4, "RAPID", 128, 2, exp_overhead=3.0)
    np.testing.assert_allclose(o2.exp_start_times, 2456196.25 + vp["exp_times"] / 1440.0, rtol=0, atol=1e-12)
    assert o2.exp_start_times[1] - o2.exp_start_times[0] == pytest.approx((obs.detector.exptime(4, 128, "RAPID") / 60.0 + 3.0) / 1440.0)


def test_example_yaml_of_the_reference_parses():
    ex = "/root/reference/examples"
    if not os.path.exists(ex):
        pytest.skip("reference tree not present")
    cfg = yaml.safe_load(open(os.path.join(ex, "hd209458b_12181_simulation_parameters.yml")))
    obs = run_visit.build_observation(cfg, ex)
    assert len(obs.exp_start_times) == 121 and obs.SUBARRAY == 256 and obs.sample_rate == 10
    assert obs.x_ref.shape == (121,) and obs.sky_background.shape == (121,)


@pytest.mark.gpu
def test_cli_runs_a_small_visit_and_writes_fits(tmp_path):
    import shutil
    work = str(tmp_path / "visit")
    shutil.copytree(MINI, work)
    obs = run_visit.run(["-p", os.path.join(work, "params.yml"), "--max-exposures", "3"])
    files = sorted(os.listdir(obs.outdir))
    assert files == ["0000_flt.fits", "0001_raw.fits", "0002_raw.fits", "0003_raw.fits", "params.yml", "visit_plan.txt"]
    h = fitsio.read(os.path.join(obs.outdir, "0002_raw.fits"))
    assert len(h) == 1 + 5 * 4 and h[0].header["NSAMP"] == 4 and h[0].header["SCAN"] is True
    # the reference's primary-header keywords and their units (exposure.py:216-410)
    p0 = h[0].header
    for key in ("DATE", "FILENAME", "FILETYPE", "TELESCOP", "INSTRUME", "EQUINOX", "PRIMESI", "TARGNAME", "RA_TARG",
                "DEC_TARG", "DATE-OBS", "TIME-OBS", "EXPSTART", "EXPEND", "EXPTIME", "POSTARG1", "POSTARG2", "OBSTYPE",
                "OBSMODE", "SCLAMP", "SUBARRAY", "SUBTYPE", "DETECTOR", "FILTER", "SAMP_SEQ", "NSAMP", "SAMPZERO",
                "APERTURE", "PROPAPER", "DIRIMAGE", "SIM", "SIM-VER", "SIM-TIME", "X-REF", "Y-REF", "SAMPRATE", "NSE-MEAN",
                "NSE-STD", "ADD-DRK", "ADD-FLAT", "ADD-GAIN", "ADD-NLIN", "STAR-NSE", "CSMCRATE", "SKY-LVL", "VSTTREND",
                "CLIPVALS", "RANDSEED", "V-PY", "V-NP", "MID-TRAN", "PERIOD", "SMA", "INC", "ECC", "PERI", "LD1", "LD4",
                "STARX"):
        assert key in p0, key
    assert p0["TARGNAME"] == "HD 209458 b" and p0["FILTER"] == "G141" and p0["SUBTYPE"] == "SQ128SUB"
    assert abs(p0["EXPSTART"] - (obs.exp_start_times[1] - 2400000.5)) < 1e-9          # Modified Julian Date
    assert abs((p0["EXPEND"] - p0["EXPSTART"]) * 86400.0 - p0["EXPTIME"]) < 1e-3
    assert p0["SAMPRATE"] == pytest.approx(0.025) and p0["RANDSEED"] == 1963 and p0["APERTURE"] == "GRISM128"
    assert p0["PERIOD"] == pytest.approx(3.524746) and p0["LD1"] == pytest.approx(0.800627)
    sci = [x for x in h if x.name == "SCI"]
    assert [x.header["SAMPNUM"] for x in sci] == [3, 2, 1, 0] and sci[0].data.shape == (138, 138)
    assert sci[0].header["SAMPTIME"] == pytest.approx(detector.WFC3_IR().exptime(4, 128, "RAPID"))
    assert sci[0].data.max() > sci[-1].data.max() - 1e4                    # last read holds the most flux... (zero read has the bias)
    # regenerate exposure 2 directly: identical (counter-based RNG, no dependence on run order)
    frame = obs._generate_exposure(obs.exp_start_times[1], 2, write_fits=False)
    np.testing.assert_array_equal(np.asarray(frame.reads[3][0], dtype=np.float64), sci[0].data)
    di = fitsio.read(os.path.join(obs.outdir, "0000_flt.fits"))
    assert abs(di[1].data.max() - 10000.0) < 700                           # the 2-D gaussian direct image


@pytest.mark.gpu
def test_cli_resume_on_the_device(tmp_path):
    # `--resume` with the real context (tests/test_resume.py has the rank-failure scenario on the CPU): a visit with one
    # file missing and one truncated is completed by regenerating exactly those two, and what is regenerated equals,
    # read for read and bit for bit, what the first run wrote -- every random stream is keyed by (visit seed, exposure
    # index), never by what ran before (the reference's exposures share one global numpy stream: no restart there)
    import shutil
    work = str(tmp_path / "visit")
    shutil.copytree(MINI, work)
    yml = os.path.join(work, "params.yml")
    obs = run_visit.run(["-p", yml, "--max-exposures", "4"])
    names = ["%04d_raw.fits" % n for n in (1, 2, 3, 4)]
    first = {n: [h.data for h in fitsio.read(os.path.join(obs.outdir, n))] for n in names}
    stamp = {n: os.stat(os.path.join(obs.outdir, n)).st_mtime_ns for n in names}
    os.remove(os.path.join(obs.outdir, names[1]))
    with open(os.path.join(obs.outdir, names[2]), "r+b") as f:
        f.truncate(100000)
    obs2 = run_visit.run(["-p", yml, "--max-exposures", "4", "--resume"])
    assert obs2.skipped == [0, 3]
    assert sorted(n for n in os.listdir(obs2.outdir) if n.endswith((".fits", ".part"))) == ["0000_flt.fits"] + names
    for n in names:
        again = [h.data for h in fitsio.read(os.path.join(obs2.outdir, n))]
        assert len(again) == len(first[n]) == 1 + 5 * 4
        for a, b in zip(again, first[n]):
            assert (a is None and b is None) or np.array_equal(a, b)
        assert (os.stat(os.path.join(obs2.outdir, n)).st_mtime_ns == stamp[n]) == (n in (names[0], names[3]))
    from wayne_amd import engine
    engine.close_all()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["gpus_2_sharing_device_0", "ranks_per_gpu_2"])
def test_cli_starts_its_own_ranks(tmp_path, how):
    # `python -m wayne_amd.run_visit --gpus 2`: the parent starts two rank processes before anything touches a GPU
    # (here both on device 0: WAYNE_SHARE_GPU=1) -- or `--gpus 1 --ranks-per-gpu 2`, two ranks to the one GPU, the
    # form for small sub-arrays; each writes its round-robin share; every file equals the one a single process writes
    import shutil
    import subprocess
    import sys
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    shutil.copytree(MINI, one)
    shutil.copytree(MINI, two)
    obs = run_visit.run(["-p", os.path.join(one, "params.yml"), "--max-exposures", "4"])
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "WAYNE_SHARE_GPU"):
        env.pop(k, None)
    if how == "gpus_2_sharing_device_0":
        env["WAYNE_SHARE_GPU"] = "1"
        ranks = ["--gpus", "2"]
    else:
        ranks = ["--gpus", "1", "--ranks-per-gpu", "2"]
    out = subprocess.run([sys.executable, "-m", "wayne_amd.run_visit", "-p", os.path.join(two, "params.yml"),
                          "--max-exposures", "4"] + ranks, capture_output=True, text=True, timeout=900, env=env,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "2 ranks done" in out.stdout
    outdir2 = os.path.join(two, os.path.relpath(obs.outdir, one))
    names = sorted(f for f in os.listdir(obs.outdir) if f.endswith("_raw.fits"))
    assert names == ["%04d_raw.fits" % i for i in range(1, 5)]
    for n in names:
        a, b = fitsio.read(os.path.join(obs.outdir, n)), fitsio.read(os.path.join(outdir2, n))
        for ha, hb in zip(a, b):
            if ha.data is not None:
                np.testing.assert_array_equal(ha.data, hb.data)
    # ... and the two ranks restarted with --resume after rank 1 lost a file: only that file is written again
    stamp = {n: os.stat(os.path.join(outdir2, n)).st_mtime_ns for n in names}
    os.remove(os.path.join(outdir2, names[1]))                  # exposure index 1: rank 1's
    out = subprocess.run([sys.executable, "-m", "wayne_amd.run_visit", "-p", os.path.join(two, "params.yml"),
                          "--max-exposures", "4", "--resume"] + ranks, capture_output=True, text=True, timeout=900, env=env,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "(2 already there)" in out.stdout and "(1 already there)" in out.stdout
    for n in names:
        assert (os.stat(os.path.join(outdir2, n)).st_mtime_ns == stamp[n]) == (n != names[1]), n
    a, b = fitsio.read(os.path.join(obs.outdir, names[1])), fitsio.read(os.path.join(outdir2, names[1]))
    for ha, hb in zip(a, b):
        if ha.data is not None:
            np.testing.assert_array_equal(ha.data, hb.data)


def test_cli_launcher_fails_loudly_without_gpus(tmp_path):
    # CPU: --gpus 2 starts two ranks; without a GPU both refuse (no CPU fallback) and the parent says so
    import shutil
    import subprocess
    import sys
    from wayne_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    work = str(tmp_path / "v")
    shutil.copytree(MINI, work)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-m", "wayne_amd.run_visit", "-p", os.path.join(work, "params.yml"),
                          "--max-exposures", "2", "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0 and "rank exit codes" in (out.stderr + out.stdout)


def test_cosmic_ray_generators_keep_the_reference_api():
    # wayne.trend_generators.cosmic_rays' public classes (cosmic_rays.py:10-139); same draws as the oracle's
    # restatement of MinMaxPossionCosmicGenerator.cosmic_frame over the same legacy stream
    from oracle import wayne_oracle as wo
    from wayne_amd.trend_generators.cosmic_rays import BaseCosmicGenerator, MinMaxPossionCosmicGenerator
    g = MinMaxPossionCosmicGenerator(11., rng=np.random.RandomState(5))
    frame = g.cosmic_frame(100.0, 256)
    want = wo.LegacyDraws(5).cosmic_frame(11., 100.0, 256, 0)
    np.testing.assert_array_equal(frame, want)
    hits = frame[frame > 0]
    assert frame.shape == (256, 256) and 30 < hits.size < 110 and hits.min() >= 10000
    assert g._rate_full_frame_to_size(11., 256) == pytest.approx(11. / 16) and g._rate_full_frame_to_size(11., (512, 1024)) == 5.5
    b = BaseCosmicGenerator(rng=np.random.RandomState(1)).cosmic_frame(2.0, (32, 64))
    assert b.shape == (32, 64) and b.sum() == 22 * 25000


def test_gpu_numa_pinning_is_best_effort():
    from wayne_amd import launch
    assert launch._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    # no such device / no sysfs entry: nothing happens, nothing raises
    assert launch.gpu_local_cpus(10 ** 6) is None
    before = os.sched_getaffinity(0)
    launch.pin_to_gpu_numa(10 ** 6)
    assert os.sched_getaffinity(0) == before


@pytest.mark.gpu
def test_cli_staring_mode_visit(tmp_path):
    # spatial_scan: False -> staring frames (one sub-sample per read) through the same pipelined driver
    import shutil
    import yaml
    work = str(tmp_path / "stare")
    shutil.copytree(MINI, work)
    pfile = os.path.join(work, "params.yml")
    cfg = yaml.safe_load(open(pfile))
    cfg["observation"]["spatial_scan"] = False
    cfg["observation"]["ssv_type"] = False
    yaml.safe_dump(cfg, open(pfile, "w"))
    obs = run_visit.run(["-p", pfile, "--max-exposures", "2"])
    files = sorted(f for f in os.listdir(obs.outdir) if f.endswith("_raw.fits"))
    assert files == ["0001_raw.fits", "0002_raw.fits"]
    h = fitsio.read(os.path.join(obs.outdir, "0001_raw.fits"))
    assert h[0].header["SCAN"] is True      # as the reference: staring_frame never resets exp_info (exposure_generator.py:162)
    sci = [x for x in h if x.name == "SCI"]
    assert len(sci) == 4
    last = sci[0].data - sci[-1].data
    # a staring spectrum: the flux sits in a few rows about the trace instead of a scanned band
    rows = last.sum(axis=1)
    assert rows.max() > 0 and (rows > 0.05 * rows.max()).sum() < 30
    # same exposure generated directly, unpipelined: identical
    frame = obs._generate_exposure(obs.exp_start_times[0], 1, write_fits=False)
    np.testing.assert_array_equal(np.asarray(frame.reads[3][0], dtype=np.float64), sci[0].data)


@pytest.mark.gpu
def test_cli_g102_visit(tmp_path):
    import shutil
    import yaml
    work = str(tmp_path / "g102")
    shutil.copytree(MINI, work)
    pfile = os.path.join(work, "params.yml")
    cfg = yaml.safe_load(open(pfile))
    cfg["observation"]["grism"] = "G102"
    yaml.safe_dump(cfg, open(pfile, "w"))
    obs = run_visit.run(["-p", pfile, "--max-exposures", "2"])
    assert obs.grism.name == "G102"
    h = fitsio.read(os.path.join(obs.outdir, "0002_raw.fits"))
    sci = [x for x in h if x.name == "SCI"]
    flux = (sci[0].data - sci[-1].data)[5:-5, 5:-5]
    assert flux.sum() > 1e4                                   # the 0.8-1.15 micron spectrum landed on the sub-array
    cols = np.nonzero(flux.sum(axis=0) > 0.02 * flux.sum(axis=0).max())[0]
    assert cols.size > 20


@pytest.mark.gpu
def test_pipelined_visit_runner_matches_direct_calls(tmp_path):
    import helpers
    from wayne_amd import visit as wv
    v = helpers.make_visit("small256", n_exposures=5)
    runner = wv.VisitRunner(v, 0, out_dir=str(tmp_path))
    seen = []
    got = runner.run(wv.shard(5, 1, 2) + wv.shard(5, 0, 2), keep=True, on_reads=lambda i, r: seen.append((i, float(r[-1].max()))))
    assert sorted(got) == [0, 1, 2, 3, 4] and [i for i, _ in seen] == [1, 3, 0, 2, 4]
    for i in (0, 3):
        direct = np.stack([r[0] for r in helpers.product_generator(v, i).scanning_frame(out_dtype=np.float32, **v.frame_kwargs(i)).reads])
        np.testing.assert_array_equal(got[i], direct)          # order of generation and slot / stream do not matter
        h = fitsio.read(os.path.join(str(tmp_path), "%04d_raw.fits" % (i + 1)))
        np.testing.assert_array_equal(h[1].data, direct[-1].astype(np.float64))


def test_ssv_modulated_sine_preserves_read_times():
    from wayne_amd.trend_generators.scan_speed_varations import SSVModulatedSine, SSVSine
    det = detector.WFC3_IR()
    rt = det.get_read_times(5, 256, "SPARS10")
    for seed in (1, 2, 3):
        g = SSVModulatedSine(10, 1.1, 100, rng_seed=seed)
        d, br = g.get_subsample_exposure_times(None, None, rt, 10.0)
        assert len(d) == 2232 and len(br) == 4 and br[-1] == 2231          # one sample fewer than the 2233 mid-points
        assert abs(d.sum() / 1000 - rt[-1]) < 2e-6
        for b, t in zip(br, rt):
            assert abs(d[:b + 1].sum() / 1000 - t) < 2e-6                   # every read lands on its table time
        assert 6.5 < d.min() and d.max() < 13.5                            # 10 ms +- 10 % x (1 + slow sine + blip)
        d2, br2 = SSVModulatedSine(10, 1.1, 100, rng_seed=seed).get_subsample_exposure_times(None, None, rt, 10.0)
        np.testing.assert_array_equal(d, d2)
    with pytest.raises(ValueError):
        SSVSine(1.5, 1.1, "rand")                                           # broken in the reference (:49)
    # through the exposure generator: host descriptor only (no GPU)
    import helpers
    from wayne_amd import visit as wv
    v = helpers.make_visit("cfg1", n_exposures=4)
    gen = wv.VisitRunner(v).generator(3)
    kw = v.frame_kwargs(3, ssv_generator=SSVModulatedSine(10, 1.1, 1))
    desc = gen.build_descriptor(None, **kw)
    dur = np.ctypeslib.as_array(desc.dur_ms, shape=(desc.n_samples,))
    sread = np.ctypeslib.as_array(desc.sample_read, shape=(desc.n_samples,))
    assert desc.n_samples == 2233 and dur[-1] == 0.0 and abs(dur.sum() / 1000 - rt[-1]) < 2e-6
    assert np.all(np.diff(sread) >= 0) and sread[0] == 0 and sread[-1] == 3
    for r in range(4):
        assert abs(dur[sread <= r].sum() / 1000 - rt[r]) < 2e-6


@pytest.mark.parametrize("mode", [(5, 256, "SPARS10", 10.0), (16, 1024, "SPARS10", 40.0), (8, 512, "SPARS25", 25.0),
                                  (4, 64, "RAPID", 1.0)])
def test_ssv_modulated_sine_against_the_oracle(mode):
    # the product's generator against oracle/wayne_oracle.py's statement-for-statement restatement of
    # scan_speed_varations.py:63-171, both over a numpy legacy stream with the same seed: same draws in the same
    # order -> the same microsecond bookkeeping -> equal durations and read indexes
    from oracle import wayne_oracle as wo
    from wayne_amd.trend_generators.scan_speed_varations import SSVModulatedSine
    nsamp, sub, seq, rate = mode
    rt = detector.WFC3_IR().get_read_times(nsamp, sub, seq)
    for seed in (0, 1, 7, 2024):
        for blip in (0, 100):
            got_d, got_i = SSVModulatedSine(10, 1.1, blip, rng_seed=seed).get_subsample_exposure_times(None, None, rt, rate)
            want_d, want_i = wo.SSVModulatedSine(10, 1.1, blip).get_subsample_exposure_times(
                None, None, rt, rate, rs=np.random.RandomState(seed))
            assert list(got_i) == list(want_i), (seed, blip)
            np.testing.assert_array_equal(got_d, want_d)
            assert len(want_i) == nsamp - 1 and abs(want_d.sum() / 1000 - rt[-1]) < 2e-6


@pytest.mark.gpu
def test_exposure_with_modulated_sine_ssv():
    import helpers
    from wayne_amd.trend_generators.scan_speed_varations import SSVModulatedSine
    v = helpers.make_visit("cfg1")
    pg = helpers.product_generator(v, 0)
    quiet = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False,
                 add_flat=False, add_gain_variations=False, add_non_linear=False, clip_values_det_limits=False,
                 add_initial_bias=False)
    rec = {}
    reads = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float64, record=rec,
                                                      **v.frame_kwargs(0, ssv_generator=SSVModulatedSine(10, 1.1, 100),
                                                                       **quiet)).reads])
    assert rec["dur"].shape == (2233,) and abs(rec["dur"].sum() / 1000 - v.read_times[-1]) < 2e-6
    np.testing.assert_allclose(reads[-1].sum() * 2.35, rec["acc"].sum(), rtol=1e-12)
    assert np.all(np.diff(reads, axis=0) >= -1e-9)
    # the electrons of a read interval scale with its (unchanged) length
    per_read = rec["acc"].reshape(4, -1).sum(axis=1)
    dt = np.diff(np.concatenate([[0.0], v.read_times]))
    np.testing.assert_allclose(per_read / per_read.sum(), dt / dt.sum(), atol=0.01)


@pytest.mark.gpu
@pytest.mark.parametrize("blip", [0, 100])
def test_modulated_sine_exposure_against_the_oracle(blip):
    # SURVEY section 8(f4), scan_speed_varations.py:63-171 through a whole exposure: the reference's example-visit
    # shape (cfg1: 256 x 256, NSAMP 5, 10 ms sampling -> 2233 sub-samples) driven by SSVModulatedSine, through the HIP
    # path and through ExposureOracle (whose restatement of the generator draws from a numpy legacy stream with the
    # product's per-exposure key; its read indexes trigger the reads as `if i in read_index` does,
    # exposure_generator.py:361).  Deterministic switches, replay thrower: durations and read indexes equal, counts per
    # bin exact (fp64 samplers on both sides: at most two of the 1e7 draws may fall on a 1-ulp boundary, and are then
    # accounted for), positions 1e-9 px, accumulated electrons to the fixed-point quantum, reads to 1e-4 DN (float64 out).
    import helpers
    from oracle import wayne_oracle as wo
    from wayne_amd import _lib
    from wayne_amd.trend_generators.scan_speed_varations import SSVModulatedSine
    i = 2
    v = helpers.make_visit("cfg1", n_exposures=i + 1)
    pg = helpers.product_generator(v, i)
    # (stellar Poisson noise stays on: the expected count of a bin in a 10 ms sub-sample is 2.5 electrons -- np.round
    # would leave a few ones in a frame of zeros; the draws are the same Philox counters on both sides)
    det_off = dict(sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False)
    kw = v.frame_kwargs(i, ssv_generator=SSVModulatedSine(10, 1.1, blip), **det_off)
    rec = {}
    got = np.stack([r[0] for r in pg.scanning_frame(threads=2, rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                    record=rec, exact_samplers=True, **kw).reads])
    eo = helpers.oracle_generator(v)
    orec = {}
    okw = helpers.oracle_kwargs(kw, seed=v.seed, exposure=i)
    assert isinstance(okw["ssv_generator"], wo.SSVModulatedSine)
    want = np.stack(eo.scanning_frame(threads=2, draws=wo.PhiloxDraws(v.seed, i, 256), thrower="oracle", record=orec,
                                      **okw))
    # the generator's own output, product against oracle, on the stream of this exposure
    key = (v.seed * 1000003 + i * 7919 + 12345) & 0x7FFFFFFF
    d_or, idx_or = wo.SSVModulatedSine(10, 1.1, blip).get_subsample_exposure_times(
        None, None, eo.read_times, 10.0, rs=np.random.RandomState(key))
    K = rec["dur"].size
    assert K == 2233 and len(d_or) in (K, K - 1)
    np.testing.assert_array_equal(rec["dur"][:len(d_or)], d_or)
    assert not rec["dur"][len(d_or):].any()                         # a sample without a duration exposes for 0 ms (:337-342)
    # the sub-sample that closes each read but the last (`if i in read_index`, :361)
    assert [int(k) for k in np.nonzero(np.diff(rec["read"]))[0]] == [min(int(b), K - 1) for b in idx_or[:-1]]
    # the exposure
    oc = np.stack(orec["counts"])
    flipped = int((rec["counts"] != oc).sum())
    slack = float(np.abs(rec["counts"].astype(np.int64) - oc).sum())      # electrons of draws that fell on a boundary
    assert flipped <= 2 and oc.sum() > 2e7
    np.testing.assert_allclose(rec["x"], np.stack(orec["x"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(rec["y"], np.stack(orec["y"]), rtol=0, atol=1e-9)
    acc_o = np.stack(orec["acc"])
    d_acc = np.abs(rec["acc"] - acc_o)
    assert d_acc.sum() <= 2.0 * slack + acc_o.size * 2233 * 2.0 ** -29
    assert d_acc.max() <= slack + 2233 * 2.0 ** -29 + 1e-9
    assert np.abs(got - want).max() <= 1e-4 + slack / 2.0
    assert np.abs(got[-1]).max() > 50


def test_cli_launcher_at_eight_ranks_dry_run(tmp_path):
    # CPU: `python -m wayne_amd.run_visit -p ... --gpus 8 --dry-run` -- the CLI's own launcher at the rank count of an
    # 8-GPU node: eight fresh rank processes, each with its RANK / LOCAL_RANK / WORLD_SIZE, its device = its local rank,
    # a capped host thread count, and its round-robin share of the visit; together they cover every exposure once.
    # (bench.py's launcher is rehearsed at N = 8 by tests/test_bench_contract.py; no GPU work in either.)
    import subprocess
    import sys
    pfile = os.path.join(MINI, [f for f in os.listdir(MINI) if f.endswith(".yml")][0])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OMP_NUM_THREADS")}
    env["PYTHONPATH"] = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, "-m", "wayne_amd.run_visit", "-p", pfile, "--gpus", "8", "--dry-run",
                          "--max-exposures", "21"], capture_output=True, text=True, timeout=600, env=env,
                         cwd=os.path.dirname(HERE))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = sorted(l for l in out.stdout.splitlines() if l.startswith("dry-run rank"))
    assert len(lines) == 8 and "run_visit: 8 ranks done" in out.stdout
    seen = []
    from wayne_amd import launch
    want_threads = str(launch.host_threads_per_rank(8))
    for r, line in enumerate(lines):
        f = line.split()
        assert f[2] == "%d/8" % r and f[4] == str(r) and f[6] == want_threads, line
        seen += [int(i) for i in f[8].split(",")]
    assert sorted(seen) == list(range(21))
    # --device with several ranks would put them all on one GPU: refused
    bad = subprocess.run([sys.executable, "-m", "wayne_amd.run_visit", "-p", pfile, "--gpus", "2", "--device", "0",
                          "--dry-run"], capture_output=True, text=True, timeout=120, env=env, cwd=os.path.dirname(HERE))
    assert bad.returncode != 0 and "--device cannot be combined" in (bad.stderr + bad.stdout)


def test_rank_environment_caps_host_threads():
    from wayne_amd import launch
    env = launch.rank_env(3, 8, 12345, {"X": "1"})
    assert env["RANK"] == env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["MASTER_ADDR"] == "127.0.0.1"
    n = launch.host_threads_per_rank(8)
    assert n >= 1 and n <= max(1, (os.cpu_count() or 1))
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        assert env[var] == os.environ.get(var, str(n))


def test_fits_write_continues_after_short_writes(tmp_path, monkeypatch):
    # os.writev may write fewer bytes than it was handed (a signal, a quota, a payload beyond 2 GiB) and Python does not
    # retry: the writer has to go on from where the call stopped, across piece boundaries, until every byte is out.
    # Here every call writes at most 1000 bytes -- less than a header block -- and the file must still equal the one
    # written in one piece, for the HDU-list writer and for the exposure writer's pre-rendered pieces.
    import os as _os
    rng = np.random.default_rng(5)
    hdus = [fitsio.HDU(fitsio.Header([("OBJECT", "x", "")]), None),
            fitsio.HDU(fitsio.Header([("EXTVER", 1, "")]), rng.normal(size=(37, 53)).astype(np.float32), name="SCI"),
            fitsio.HDU(fitsio.Header([("EXTVER", 1, "")]), None, name="ERR"),
            fitsio.HDU(fitsio.Header([("EXTVER", 2, "")]), rng.integers(0, 100, (5, 7)).astype(np.int16), name="DQ")]
    whole = str(tmp_path / "whole.fits")
    fitsio.write(whole, hdus)
    real = _os.writev
    calls = []

    def short(fd, bufs):
        first = bytes(memoryview(bufs[0]).cast("B")[:1000])          # at most 1000 bytes of the first piece
        calls.append(len(first))
        return real(fd, [first])

    monkeypatch.setattr(_os, "writev", short)
    piecewise = str(tmp_path / "piecewise.fits")
    fitsio.write(piecewise, hdus)
    monkeypatch.undo()
    assert len(calls) > 10 and open(piecewise, "rb").read() == open(whole, "rb").read()
    back = fitsio.read(piecewise)
    np.testing.assert_array_equal(back[1].data, hdus[1].data)
