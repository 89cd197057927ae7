"""GPU: a short fixed-seed run of scripts/soak.py -- many exposures through the VisitRunner pipeline (slots in
rotation over the context's two streams, pinned staging and fetch buffers reused again and again), then a sample of
them regenerated one at a time and compared bit for bit."""
import numpy as np
import pytest

import helpers
from wayne_amd import _lib, visit
from wayne_amd.exposure_generator import ExposureGenerator

pytestmark = pytest.mark.gpu


def test_pipelined_visit_equals_one_at_a_time_generation():
    n = 160
    v = helpers.make_visit("cfg3", n_exposures=n)
    runner = visit.VisitRunner(v, 0)
    seen = {}
    runner.run(range(n), on_reads=lambda i, r: seen.__setitem__(i, (float(r[-1].sum()), float(r[1].max()),
                                                                    r[-1][::7, ::5].copy())))
    assert len(seen) == n and all(np.isfinite(s[0]) for s in seen.values())
    for i in sorted(set([0, 1, 2, n // 3, n // 2, n - 2, n - 1] + [int(j) for j in np.random.default_rng(1).integers(0, n, 6)])):
        eg = ExposureGenerator(v.detector, v.grism, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=v.calibration,
                               seed=v.seed, exposure_index=i)
        reads = np.stack([r[0] for r in eg.scanning_frame(out_dtype=np.float32, **v.frame_kwargs(i)).reads])
        assert float(reads[-1].sum()) == seen[i][0] and float(reads[1].max()) == seen[i][1], i
        np.testing.assert_array_equal(reads[-1][::7, ::5], seen[i][2])


def test_a_bin_beyond_the_lanes_reach_reruns_with_k_throw(monkeypatch):
    # the default launch has no k_throw (the host expects no bin beyond a lane's cap) and the lanes then take bins of
    # up to 64 x 4096 one-by-one electrons; a bin beyond that flags the run (status bit 1) and the exposure is
    # repeated with k_throw when its status is read.  With the reach lowered to 5 electrons (a test knob) nearly every
    # bin of this exposure is "beyond": the repeated run routes them as an ordinary launch does (lanes up to 4096),
    # so the frame must equal the ordinary one bit for bit -- nothing lost, nothing thrown twice
    v = helpers.make_visit("small256")
    kw = v.frame_kwargs(0)
    pg = helpers.product_generator(v, 0)
    want = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    _lib.set_knob_all("lane_reach", "5")
    rec = {}
    via_record = np.stack([r[0] for r in pg.scanning_frame(record=rec, **kw).reads])     # debug_fetch re-runs the front half
    via_download = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])              # download re-runs the exposure
    from wayne_amd import engine
    eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    desc = pg.build_descriptor(eng, **kw)
    eng.ctx.upload(3, desc)
    eng.ctx.run(3)
    eng.ctx.fetch_async(3)
    via_wait = np.array(eng.ctx.wait(3))                                                 # the pipelined path re-runs too
    # a throughput loop: run() calls only, then ONE synchronize -- which settles every slot it finds flagged
    # (wayne_ctx_synchronize), so that the reads in HBM are complete without any download having looked at them
    n0 = eng.ctx.reruns
    for slot in (4, 6, 7):
        eng.ctx.upload(slot, desc)
        eng.ctx.run(slot)
    eng.ctx.synchronize()
    assert eng.ctx.reruns == n0 + 3 and [eng.ctx.status(s) for s in (4, 6, 7)] == [0, 0, 0]
    n1 = eng.ctx.reruns
    via_sync = [eng.ctx.download(slot) for slot in (4, 6, 7)]
    assert eng.ctx.reruns == n1                                                          # nothing left for the downloads to repair
    # the blocking form: complete when it returns
    eng.ctx.upload(9, desc)
    eng.ctx.run_checked(9)
    assert eng.ctx.status(9) == 0 and eng.ctx.reruns == n1 + 1
    via_checked = eng.ctx.download(9)
    # and a slot that has been repaired once runs the general sequence from then on (no second run per launch)
    eng.ctx.run(9)
    eng.ctx.synchronize()
    assert eng.ctx.reruns == n1 + 1
    _lib.set_knob_all("lane_reach", None)
    assert (rec["counts"] * 0.2 > 5).mean() > 0.5                                        # most bins were beyond the reach
    np.testing.assert_array_equal(via_record, want)
    np.testing.assert_array_equal(via_download, want)
    np.testing.assert_array_equal(via_wait, want)
    for got in via_sync + [via_checked]:
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("name", ["small256", "tiny"])
def test_frames_do_not_depend_on_batches_or_the_thin_flush(name, monkeypatch):
    # k_prep_sub / k_lane / k_narrow take `kb` consecutive sub-samples per workgroup (chosen by the host from K), and
    # k_lane flushes from a first-touch list when it expects few electrons: launch geometry only -- same streams, same
    # integer sums -- so the reads must not change by a bit
    v = helpers.make_visit(name)
    kw = v.frame_kwargs(0)
    pg = helpers.product_generator(v, 0)
    want = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    for batch, thin in (("1", "0"), ("1", "1"), ("4", "1"), ("4", "0"), ("32", "1")):
        _lib.set_knob_all("batch", batch)
        _lib.set_knob_all("thin", thin)
        got = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
        np.testing.assert_array_equal(got, want, err_msg="WAYNE_BATCH=%s WAYNE_THIN=%s" % (batch, thin))
    # the per-electron and replay throwers go through the batched k_prep_sub too
    _lib.set_knob_all("batch", None)
    _lib.set_knob_all("thin", None)
    for mode in (_lib.RNG_PHILOX, _lib.RNG_REPLAY):
        a = np.stack([r[0] for r in pg.scanning_frame(rng_mode=mode, **kw).reads])
        _lib.set_knob_all("batch", "5")
        b = np.stack([r[0] for r in pg.scanning_frame(rng_mode=mode, **kw).reads])
        _lib.set_knob_all("batch", None)
        np.testing.assert_array_equal(a, b)


def test_thin_exposure_without_k_narrow_and_its_rerun(monkeypatch):
    # a few electrons per bin and sub-sample: the host leaves k_narrow out of the launch sequence (no bin is expected
    # to reach the multinomial's 32 narrow electrons); same frame as with the kernel launched and finding nothing
    v = helpers.make_visit("tiny")
    kw = v.frame_kwargs(0, scale_factor=0.4)
    pg = helpers.product_generator(v, 0)
    rec = {}
    a = np.stack([r[0] for r in pg.scanning_frame(record=rec, **kw).reads])
    assert rec["counts"].max() < 32 and rec["counts"].mean() > 1
    _lib.set_knob_all("keep_narrow", "1")
    b = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    _lib.set_knob_all("keep_narrow", None)
    np.testing.assert_array_equal(a, b)
    # ... and a bin that defies the estimate (the host's estimate does not know the transit-depth matrix: a depth of
    # -30 makes one bin 31 times brighter): the run flags itself and is repeated with k_narrow -- same frame as when
    # the kernel is launched from the start
    sig = np.array(kw["planet_signal"], dtype=float)
    i0 = int(np.searchsorted(v.wl, 1.3))
    sig[:, i0] = -30.0
    kw2 = dict(kw, planet_signal=sig)
    rec = {}
    c = np.stack([r[0] for r in pg.scanning_frame(record=rec, **kw2).reads])
    assert rec["counts"].max() > 60                                   # that bin went to the multinomial
    c2 = np.stack([r[0] for r in pg.scanning_frame(**kw2).reads])
    _lib.set_knob_all("keep_narrow", "1")
    d = np.stack([r[0] for r in pg.scanning_frame(**kw2).reads])
    _lib.set_knob_all("keep_narrow", None)
    np.testing.assert_array_equal(c, d)
    np.testing.assert_array_equal(c2, d)
    assert rec["acc"].sum() > 0.4 * rec["counts"].sum()            # (a 64-px frame: much of the scan falls off it)


@pytest.mark.parametrize("name,scale", [("tiny", 0.4), ("small256", 0.02), ("cfg1", 1.0)])
def test_fused_thin_path_equals_the_three_kernel_path(name, scale, monkeypatch):
    # thin exposures (a few electrons per bin and sub-sample, nothing expected for k_narrow or k_throw) run without
    # k_prep_sub: k_lane<.., FUSED> plans each bin itself (plan_bin, k_prep.h) and throws at once.  Same counts, same
    # positions (worked out afterwards by k_prep_sub for the record), same cosmic rays, same reads as the
    # k_prep_sub -> k_lane sequence, bit for bit; cfg1 (K = 2233 sub-samples of ~2.5 e- per bin) is the case it is for
    v = helpers.make_visit(name)
    kw = v.frame_kwargs(0, scale_factor=scale)
    pg = helpers.product_generator(v, 0)
    rec_f, rec_u = {}, {}
    a = np.stack([r[0] for r in pg.scanning_frame(record=rec_f, **kw).reads])
    a2 = np.stack([r[0] for r in pg.scanning_frame(**kw).reads])
    _lib.set_knob_all("no_fuse", "1")
    b = np.stack([r[0] for r in pg.scanning_frame(record=rec_u, **kw).reads])
    _lib.set_knob_all("no_fuse", None)
    assert rec_u["counts"].max() < 32 and rec_u["counts"].sum() > 1000
    for key in ("counts", "x", "y", "acc"):
        np.testing.assert_array_equal(rec_f[key], rec_u[key], err_msg=key)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a2, b)


@pytest.mark.parametrize("case", ["nan_and_negative_flux", "off_the_frame", "one_bright_bin"])
def test_fused_thin_path_on_hostile_inputs(case, monkeypatch):
    # the fused kernel sizes its LDS tile before it knows the counts: inputs that leave the sane range -- NaN and
    # negative flux (no electrons), a reference position far off the frame (no tile at all: every electron takes the
    # bounds-checked path and is dropped), one bin a thousand times brighter than the host's estimate can know
    # (status bit 1: the exposure is run again with k_prep_sub, k_narrow and k_throw) -- must come out as from
    # the three-kernel sequence, bit for bit
    v = helpers.make_visit("tiny")
    kw = v.frame_kwargs(0, scale_factor=0.4)
    if case == "nan_and_negative_flux":
        flux = np.array(kw["stellar_flux"], dtype=float)
        flux[::7] = np.nan
        flux[3::11] = -flux[3::11]
        kw = dict(kw, stellar_flux=flux)
    elif case == "off_the_frame":
        kw = dict(kw, x_ref=kw["x_ref"] + 3000.0, y_ref=kw["y_ref"] - 2500.0)
    else:
        sig = np.array(kw["planet_signal"], dtype=float)
        sig[:, sig.shape[1] // 2] = -2000.0
        kw = dict(kw, planet_signal=sig)
    pg = helpers.product_generator(v, 0)
    rec_f, rec_u = {}, {}
    a = np.stack([r[0] for r in pg.scanning_frame(record=rec_f, **kw).reads])
    _lib.set_knob_all("no_fuse", "1")
    b = np.stack([r[0] for r in pg.scanning_frame(record=rec_u, **kw).reads])
    _lib.set_knob_all("no_fuse", None)
    for key in ("counts", "acc"):
        np.testing.assert_array_equal(rec_f[key], rec_u[key], err_msg=key)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all()
    if case == "off_the_frame":
        assert rec_f["acc"].sum() == 0 and rec_f["counts"].sum() > 1000
    if case == "one_bright_bin":
        assert rec_f["counts"].max() > 1000
    if case == "nan_and_negative_flux":
        assert (rec_f["counts"][:, ::7] == 0).all() and rec_f["counts"].sum() > 1000


@pytest.mark.parametrize("name,kw_over", [("small256", {}), ("tiny_g102", {}), ("cfg5", {"E": 2e6, "K": 16}),
                                          ("stare256", {})])
def test_accumulator_boxes_lose_nothing(name, kw_over, monkeypatch):
    # k_ramp loads a read's accumulators only inside the host's bound on where that read's electrons can land and
    # where a cosmic-ray segment bit is set; with the boxes off (every accumulator loaded, as in rounds 1-2) the
    # reads must be the same bit for bit -- cosmic rays, jitter, SSV, scan and stare, every rng mode
    v = helpers.make_visit(name, **kw_over)
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0)
    staring = v.scan_speed == 0
    if staring:
        kw = {k: kw[k] for k in kw if k not in ("scan_speed", "sample_rate", "ssv_generator")}
    gen = pg.staring_frame if staring else pg.scanning_frame
    for mode in (_lib.RNG_SPLIT, _lib.RNG_PHILOX, _lib.RNG_REPLAY):
        a = np.stack([r[0] for r in gen(rng_mode=mode, **kw).reads])
        _lib.set_knob_all("no_acc_box", "1")
        b = np.stack([r[0] for r in gen(rng_mode=mode, **kw).reads])
        _lib.set_knob_all("no_acc_box", None)
        np.testing.assert_array_equal(a, b, err_msg="%s rng_mode %d" % (name, mode))
        # and a second exposure in the same slot starts from clean accumulators and segment bits
        c = np.stack([r[0] for r in gen(rng_mode=mode, **kw).reads])
        np.testing.assert_array_equal(a, c)


def test_accumulator_boxes_with_wavelengths_out_of_order(monkeypatch):
    # the C ABI does not ask for increasing wavelengths: a descriptor whose array starts in the middle of the band (the
    # grid rotated by half its length: first and last element are neighbours in wavelength, the bins between them cover
    # the whole trace) must put every electron inside the boxes k_ramp loads -- same reads as with the boxes off, and
    # accumulators left clean for the next exposure on the slot
    from wayne_amd import _lib, engine
    for mode in (_lib.RNG_SPLIT, _lib.RNG_PHILOX):
        v = helpers.make_visit("small256")       # (a fresh one per pass: the descriptor's arrays are views of the visit's)
        pg = helpers.product_generator(v, 0)
        eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
        kw = v.frame_kwargs(0, cosmic_rate=None)
        desc = pg.build_descriptor(eng, rng_mode=mode, out_dtype=np.float64, **kw)
        wl, flux = desc._keep[0], desc._keep[1]
        W = wl.size
        assert wl.shape == flux.shape == (W,) and np.all(np.diff(wl) > 0)
        shift = W // 2
        wl[:] = np.roll(wl, shift)
        flux[:] = np.roll(flux, shift)
        for a in desc._keep:
            if getattr(a, "shape", None) == (v.K, W):
                a[:] = np.roll(a, shift, axis=1)
        assert abs(wl[0] - wl[-1]) < 1e-3 and wl.max() - wl.min() > 0.5
        ctx = eng.ctx
        a = ctx.synthesize(desc)
        _, _, _, acc_after = ctx.debug_fetch(0, acc=True)
        assert not acc_after.any(), "accumulators left dirty: %g electrons" % acc_after.sum()
        _lib.set_knob_all("no_acc_box", "1")
        b = ctx.synthesize(desc)
        _lib.set_knob_all("no_acc_box", None)
        np.testing.assert_array_equal(a, b, err_msg="rng_mode %d" % mode)
        assert (a[-1] - a[0]).max() > 50              # the star is there


def test_a_new_grism_on_a_live_context_forgets_the_old_spectrum_estimates():
    # wayne_ctx_set_grism on a context that has already synthesized exposures: what the upload keeps per spectrum
    # (count rates per bin through the OLD sensitivity, which size the throwers' launches; the OLD largest PSF sigma,
    # which sizes the boxes of the accumulators k_ramp loads) must not survive -- the same descriptor through a context
    # whose grism was swapped for one six times as sensitive with a PSF three times as wide gives the reads of a
    # context that was built with the new grism, leaves no electron behind in the accumulators, and reruns no more
    from wayne_amd import _lib, engine
    v = helpers.make_visit("small256")
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0, cosmic_rate=None)
    sens_wl, sens_val = v.calibration.sensitivity(v.grism.name)
    g = v.grism

    def set_grism(eng, scale):
        wmin, wmax = v.calibration.flat_wl.get(g.name, (0.0, 1.0))
        eng.ctx.set_grism(g.trace_coeff, g.wl_solution, g.psf_ratio_poly.coeffs, np.asarray(g.psf_sigmal_poly.coeffs) * 3,
                          np.asarray(g.psf_sigmah_poly.coeffs) * 3, sens_wl, np.asarray(sens_val) * scale, wmin, wmax)

    for mode in (_lib.RNG_SPLIT, _lib.RNG_PHILOX):
        swapped = engine.Engine(0, g, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
        fresh = engine.Engine(0, g, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
        try:
            desc = pg.build_descriptor(swapped, rng_mode=mode, out_dtype=np.float64, **kw)
            first = swapped.ctx.synthesize(desc)
            set_grism(swapped, 6.0)
            set_grism(fresh, 6.0)
            a = swapped.ctx.synthesize(desc)
            _, _, _, acc_after = swapped.ctx.debug_fetch(0, acc=True)
            assert not acc_after.any(), "accumulators left dirty: %g electrons" % acc_after.sum()
            b = fresh.ctx.synthesize(desc)
            np.testing.assert_array_equal(a, b, err_msg="rng_mode %d" % mode)
            assert (a[-1] - a[0]).sum() > 2 * (first[-1] - first[0]).sum()      # (non-linearity and the clip eat part of the 6x)
            assert swapped.ctx.reruns == fresh.ctx.reruns
        finally:
            swapped.close()
            fresh.close()


def test_a_malformed_sensitivity_table_is_refused_and_the_context_keeps_its_grism():
    # np.interp's precondition (grism.py:116-118) is the ABI's: wayne_ctx_set_grism answers WAYNE_E_INVALID to a table with a
    # NaN / infinite entry or wavelengths that step back (tests/test_host_plan.py holds the planner's side under
    # AddressSanitizer), changes nothing, and the context goes on producing the frames of the grism it had
    from wayne_amd import _lib, engine
    v = helpers.make_visit("small256")
    pg = helpers.product_generator(v, 0)
    kw = v.frame_kwargs(0, cosmic_rate=None)
    g = v.grism
    sens_wl, sens_val = (np.asarray(a, dtype=float) for a in v.calibration.sensitivity(g.name))
    wmin, wmax = v.calibration.flat_wl.get(g.name, (0.0, 1.0))
    eng = engine.Engine(0, g, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    try:
        desc = pg.build_descriptor(eng, rng_mode=_lib.RNG_SPLIT, out_dtype=np.float32, **kw)
        before = eng.ctx.synthesize(desc)
        bad = {}
        w = sens_wl.copy(); w[-1] = np.nan; bad["nan_last"] = (w, sens_val)
        w = sens_wl.copy(); w[0] = -np.inf; bad["inf_first"] = (w, sens_val)
        bad["decreasing"] = (sens_wl[::-1].copy(), sens_val)
        w = sens_wl.copy(); w[5] = w[3]; bad["one_step_back"] = (w, sens_val)
        x = sens_val.copy(); x[len(x) // 2] = np.nan; bad["nan_value"] = (sens_wl, x)
        for name, (w, x) in bad.items():
            with pytest.raises(_lib.WayneError) as e:
                eng.ctx.set_grism(g.trace_coeff, g.wl_solution, g.psf_ratio_poly.coeffs, g.psf_sigmal_poly.coeffs,
                                  g.psf_sigmah_poly.coeffs, w, x, wmin, wmax)
            assert e.value.status == _lib.E_INVALID and "sensitivity" in str(e.value), name
            np.testing.assert_array_equal(eng.ctx.synthesize(desc), before, err_msg=name)
        # repeated wavelengths are within the precondition
        eng.ctx.set_grism(g.trace_coeff, g.wl_solution, g.psf_ratio_poly.coeffs, g.psf_sigmal_poly.coeffs,
                          g.psf_sigmah_poly.coeffs, np.repeat(sens_wl[::2], 2)[:sens_wl.size], sens_val, wmin, wmax)
        assert np.isfinite(eng.ctx.synthesize(desc)).all()
    finally:
        eng.close()


def test_random_exposures_against_the_oracle():
    # a fixed-seed stretch of scripts/soak_exposure.py: random small configuration, brightness, detector switches, sky,
    # cosmic rays, rng mode, exact / production samplers, float32 / float64 reads -- whole exposures against the numpy
    # oracle on the same counters (3000 cases of it: profiles/r04/soak.txt)
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import soak_exposure
    rng = np.random.default_rng(20261004)
    modes = set()
    for j in range(60):
        c = soak_exposure.case(rng)
        ok, bad, size, med, worst = soak_exposure.run_case(c)
        assert ok, "case %d %r: %d of %d pixels off, median %.2e, max %.2f" % (j, c, bad, size, med, worst)
        modes.add((c["mode"][0], c["exact"], c["f64"]))
    assert len(modes) >= 10          # the stretch visits nearly every (rng mode, samplers, dtype) combination


def test_not_a_number_in_the_descriptor_throws_nothing_and_corrupts_nothing():
    # hostile VALUES through the real upload path (the CPU harness of the host planner -- tests/test_host_plan.py --
    # covers the planner alone): a NaN wavelength poisons its own bin (position, PSF, sensitivity) and the widths of its
    # two neighbours; a NaN or negative flux its own bin.  The reference would cast those counts to C ints; here such a
    # bin throws nothing (k_prep.h plan_bin), the host takes the load-everything path for k_ramp (no bound can be built
    # on a NaN), and every other bin draws what it drew before: the reads equal, bit for bit, those of the clean
    # descriptor with the affected bins' flux set to zero -- and the accumulators are left clean
    from wayne_amd import _lib, engine
    for mode in (_lib.RNG_SPLIT, _lib.RNG_PHILOX, _lib.RNG_REPLAY):
        reads = []
        for hostile in (True, False):
            v = helpers.make_visit("small256")       # (a fresh one per pass: the descriptor's arrays are views of the visit's)
            pg = helpers.product_generator(v, 0)
            eng = engine.get_engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
            desc = pg.build_descriptor(eng, rng_mode=mode, out_dtype=np.float64, **v.frame_kwargs(0, cosmic_rate=None))
            wl, flux = desc._keep[0], desc._keep[1]
            W = wl.size
            j, a, b = W // 3, W // 2, (2 * W) // 3
            if hostile:
                wl[j] = np.nan
                flux[a] = np.nan
                flux[b] = -flux[b]
            else:
                flux[j - 1:j + 2] = 0.0
                flux[a] = 0.0
                flux[b] = 0.0
            ctx = eng.ctx
            reads.append(ctx.synthesize(desc))
            use_box, _, _ = ctx.debug_boxes(0)
            assert use_box == (not hostile)
            _, _, _, acc_after = ctx.debug_fetch(0, acc=True)
            assert not acc_after.any(), "accumulators left dirty: %g electrons" % acc_after.sum()
            assert ctx.status(0) == 0
        assert np.isfinite(reads[0]).all()
        np.testing.assert_array_equal(reads[0], reads[1], err_msg="rng_mode %d" % mode)
        assert (reads[0][-1] - reads[0][0]).max() > 50              # the star is there
