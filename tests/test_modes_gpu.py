"""GPU: every mode axis of the path against the oracle -- both grisms, the 64 / 128 / 256 / 512
sub-arrays (frame offset 507 - SUBARRAY/2, exposure_generator.py:630; flat offset (1014 - size)/2,
grism.py:363; linearity crop, detector.py:328-333), staring frames, float32 / float64 reads, the
SUBARRAY = 1024 quirk switch, and the error paths of the C ABI."""
import numpy as np
import pytest

import helpers
from oracle import wayne_oracle as wo
from wayne_amd import _lib, engine

pytestmark = pytest.mark.gpu

OFF = dict(add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False, add_read_noise=False)


def both(name, thrower="oracle", rng_mode=_lib.RNG_REPLAY, staring=False, **over):
    v = helpers.make_visit(name)
    kw = v.frame_kwargs(0, **over)
    pg = helpers.product_generator(v, 0)
    eo = helpers.oracle_generator(v)
    N = v.detector.light_sensitive_size(v.SUBARRAY)
    draws = wo.PhiloxDraws(v.seed, 0, N)
    rec, orec = {}, {}
    if staring:
        skw = {k: kw[k] for k in kw if k not in ("scan_speed", "sample_rate", "ssv_generator")}
        exp = pg.staring_frame(threads=2, rng_mode=rng_mode, out_dtype=np.float64, exact_samplers=True, record=rec, **skw)
        want = eo.staring_frame(threads=2, draws=draws, thrower=thrower, record=orec, **helpers.oracle_kwargs(skw))
    else:
        exp = pg.scanning_frame(threads=2, rng_mode=rng_mode, out_dtype=np.float64, exact_samplers=True, record=rec, **kw)
        want = eo.scanning_frame(threads=2, draws=draws, thrower=thrower, record=orec, **helpers.oracle_kwargs(kw))
    return v, np.stack([r[0] for r in exp.reads]), np.stack(want), rec, orec


@pytest.mark.parametrize("name", ["tiny_g102", "tiny128", "tiny512", "tiny"])
def test_deterministic_parity_all_subarrays_and_grisms(name):
    v, got, want, rec, orec = both(name, **OFF)
    np.testing.assert_array_equal(rec["counts"], np.stack(orec["counts"]))
    np.testing.assert_allclose(rec["x"], np.stack(orec["x"]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(rec["acc"], np.stack(orec["acc"]), rtol=1e-13, atol=4096.0 * v.K * 2.0 ** -29)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-4)
    assert np.stack(orec["acc"]).sum() > 1000                      # the spectrum is on the frame
    S = v.detector.full_size(v.SUBARRAY)
    assert got.shape == (v.NSAMP, S, S)


def test_g102_uses_its_own_trace_and_limits():
    v, got, want, rec, orec = both("tiny_g102", **OFF)
    v2 = helpers.make_visit("tiny")
    assert v.grism.wl_limits == (0.75, 1.2) and v.grism.trace_coeff != v2.grism.trace_coeff
    # G102 disperses ~24.5 A/px: x positions of the first / last bin span (1.2 - 0.75) um / 24.5 A
    span = rec["x"][0].max() - rec["x"][0].min()
    assert 170 < span < 195


@pytest.mark.parametrize("name", ["tiny128", "tiny_g102"])
def test_noisy_parity_exact_samplers(name):
    v, got, want, rec, orec = both(name, add_stellar_noise=True)
    d = np.abs(got - want)
    assert (d > 1e-3 + 1e-6 * np.abs(want)).sum() <= max(2, 2e-4 * got.size)
    assert np.median(d) < 1e-4


def test_staring_frame_against_oracle():
    v, got, want, rec, orec = both("stare256", staring=True, **OFF)
    np.testing.assert_array_equal(rec["counts"], np.stack(orec["counts"]))
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-4)
    assert len(rec["dur"]) == v.NSAMP - 1                           # one sample per read
    # without a scan every read interval lands on the same rows
    rows = [np.average(np.arange(a.shape[0]), weights=a.sum(axis=1)) for a in rec["acc"]]
    assert max(rows) - min(rows) < 0.05


def test_float32_reads_are_rounded_float64_reads():
    v = helpers.make_visit("tiny128")
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0)
    # exact samplers: one arithmetic, the float32 read is the float64 read rounded once
    a = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float64, exact_samplers=True, **kw).reads])
    b = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float32, exact_samplers=True, **kw).reads])
    assert a.dtype == np.float64 and b.dtype == np.float32
    np.testing.assert_array_equal(a.astype(np.float32), b)
    # production math: the float32 variant of k_ramp does its per-read arithmetic in float32 where a float32 read
    # cannot tell (k_ramp.h, "PRODUCTION VARIANT"): same draws, reads within the float32 tolerance
    a = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float64, **kw).reads])
    b = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float32, **kw).reads])
    np.testing.assert_allclose(b, a, rtol=2e-7, atol=0.02)
    assert np.abs(b - a).max() < 0.01 and np.median(np.abs(b - a)) < 1e-3


@pytest.mark.parametrize("name", ["small256", "cfg4"])
def test_production_float32_ramp_against_float64_ramp(name):
    # the same at sizes where pixels run up the non-linear part of the ramp (tens of thousands of DN), every
    # detector switch on: the production float32 k_ramp against the float64 variant in the same (hardware) math
    v = helpers.make_visit(name)
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0)
    a = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float64, **kw).reads])
    b = np.stack([r[0] for r in pg.scanning_frame(out_dtype=np.float32, **kw).reads])
    assert a[-1].max() > 5000
    d = np.abs(b - a)
    np.testing.assert_allclose(b, a, rtol=2e-7, atol=0.02)
    assert np.median(d) < 1e-3


def test_reference_quirk_switch_at_1024():
    # the reference's 507 - 1024/2 = -5 shifts the spectrum by +5 px on the 1014 frame
    v = helpers.make_visit("cfg2", E=2e6)
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0, **dict(OFF, add_flat=False, add_gain_variations=False, add_non_linear=False))
    ra, rb = {}, {}
    pg.scanning_frame(rng_mode=_lib.RNG_REPLAY, record=ra, **kw)
    pg.scanning_frame(rng_mode=_lib.RNG_REPLAY, reference_quirks=True, record=rb, **kw)
    np.testing.assert_allclose(rb["x"], ra["x"] + 5.0, atol=1e-9)
    np.testing.assert_allclose(rb["y"], ra["y"] + 5.0, atol=1e-9)
    a, b = ra["acc"].sum(axis=0), rb["acc"].sum(axis=0)
    assert abs(a.sum() - b.sum()) < 0.01 * a.sum()
    ca = np.average(np.arange(a.shape[1]), weights=a.sum(axis=0))
    cb = np.average(np.arange(b.shape[1]), weights=b.sum(axis=0))
    assert abs(cb - ca - 5.0) < 0.3


def test_g102_flat_quirk_switch():
    # reference_quirks=True flat-fields a G102 exposure with the G141 cube, as the reference does
    # (grism.py:428,453-454); the default uses the G102 cube.  Each against the oracle built the same way.
    v = helpers.make_visit("tiny_g102")
    kw = v.frame_kwargs(0, add_stellar_noise=False, sky_background=0.0, cosmic_rate=None, add_dark=False,
                        add_read_noise=False)
    pg = helpers.product_generator(v, 0)
    frames = {}
    for quirk in (False, True):
        got = np.stack([r[0] for r in pg.scanning_frame(threads=2, rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                        reference_quirks=quirk, **kw).reads])
        det, gr, eo = wo.from_calibration(v.calibration, "G102", v.NSAMP, v.SAMPSEQ, v.SUBARRAY, g102_flat_quirk=quirk)
        want = np.stack(eo.scanning_frame(threads=2, draws=wo.PhiloxDraws(v.seed, 0, 64), thrower="oracle",
                                          reference_quirks=quirk, **helpers.oracle_kwargs(kw)))
        assert np.abs(got - want).max() < 1e-4
        frames[quirk] = got
    assert np.abs(frames[True] - frames[False]).max() > 1e-3      # the two cubes differ


@pytest.mark.parametrize("case", range(10))
def test_random_switch_combinations_against_oracle(case):
    # a different configuration, number of sub-samples and set of detector switches each time: the device
    # and the oracle read the same descriptor semantics whatever the combination (bit-exact replay thrower,
    # exact samplers, Philox-keyed noise on both sides)
    rng = np.random.default_rng(1000 + case)
    name = ["tiny", "tiny128", "tiny_g102", "small256", "stare256"][case % 5]
    K = None if name in ("small256", "stare256") else int(rng.integers(1, 9))
    v = helpers.make_visit(name, **({} if K is None else {"K": max(K, v_min_k(name))}))
    over = dict(add_flat=bool(rng.integers(2)), add_gain_variations=bool(rng.integers(2)),
                add_non_linear=bool(rng.integers(2)), clip_values_det_limits=bool(rng.integers(2)),
                add_initial_bias=bool(rng.integers(2)), add_dark=bool(rng.integers(2)),
                add_read_noise=bool(rng.integers(2)), add_stellar_noise=bool(rng.integers(2)),
                sky_background=[0.0, 0.4, 7.5][int(rng.integers(3))],
                cosmic_rate=[None, 30.0][int(rng.integers(2))],
                scale_factor=float(rng.uniform(0.5, 3.0)))
    if rng.integers(2):
        over.update(noise_mean=float(rng.uniform(0.5, 3.0)), noise_std=float(rng.uniform(0.1, 1.0)))
    kw = v.frame_kwargs(0, **over)
    pg = helpers.product_generator(v, 0)
    eo = helpers.oracle_generator(v)
    N = v.detector.light_sensitive_size(v.SUBARRAY)
    got = np.stack([r[0] for r in pg.scanning_frame(threads=3, rng_mode=_lib.RNG_REPLAY, out_dtype=np.float64,
                                                    exact_samplers=True, **kw).reads])
    want = np.stack(eo.scanning_frame(threads=3, draws=wo.PhiloxDraws(v.seed, 0, N), thrower="oracle",
                                      **helpers.oracle_kwargs(kw)))
    assert got.shape == want.shape
    d = np.abs(got - want)
    bad = int((d > 1e-3 + 1e-6 * np.abs(want)).sum())
    if over["add_stellar_noise"]:
        # one differing Poisson count re-numbers the electrons of that sub-sample in the replay thrower
        assert bad <= 0.02 * got.size, (over, bad)
    else:
        assert bad <= 3e-4 * got.size, (over, bad)
    assert np.median(d) < 1e-4


def v_min_k(name):
    # at least one sub-sample per read
    return {"tiny": 3, "tiny128": 4, "tiny_g102": 2}.get(name, 1)


def test_abi_error_paths():
    v = helpers.make_visit("tiny")
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    ctx = eng.ctx
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0)

    def desc(**over):
        return pg.build_descriptor(eng, **dict(kw, **over))

    d = desc()
    d.n_reads = 2
    with pytest.raises(_lib.WayneError) as e:
        ctx.upload(0, d)
    assert e.value.status == _lib.E_INVALID and "n_reads" in str(e.value)
    d = desc()
    d._keep[5][0] = 99                                # sample_read out of range
    with pytest.raises(_lib.WayneError):
        ctx.upload(0, d)
    with pytest.raises(_lib.WayneError):
        ctx.upload(999, desc())
    with pytest.raises(_lib.WayneError) as e:
        ctx.run(7)                                     # never uploaded
    assert e.value.status == _lib.E_STATE
    d = desc(rng_mode=_lib.RNG_REPLAY)
    d.threads_compat = 0
    with pytest.raises(_lib.WayneError):
        ctx.upload(0, d)
    # too many electrons for the 32-bit electron index: loud overflow, not garbage
    ctx.upload(0, desc(stellar_flux=kw["stellar_flux"] * 1e9))
    ctx.run(0)
    with pytest.raises(_lib.WayneError) as e:
        ctx.download(0)
    assert e.value.status == _lib.E_OVERFLOW
    # the context still works afterwards
    ctx.upload(0, desc())
    ctx.run(0)
    assert np.isfinite(ctx.download(0)).all()
    # a context without calibration refuses exposures
    bare = _lib.Context(0)
    with pytest.raises(_lib.WayneError) as e:
        bare.upload(0, desc())
    assert e.value.status == _lib.E_STATE
    bare.close()
