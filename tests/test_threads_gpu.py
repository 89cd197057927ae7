"""GPU: include/wayne_hip.h promises "calls on distinct contexts are thread-safe, calls on one are not".  Two contexts on
device 0, each driven by its own Python thread through upload / run / download, psf_apply and the pipelined
fetch_async / wait of different exposures at the same time (the ctypes calls release the interpreter lock, so the two
threads really are inside the library together): every frame must equal, bit for bit, the one a single thread produces.
"""
import threading

import numpy as np
import pytest

import helpers
from conftest import load_golden_psf
from wayne_amd import _lib, engine

pytestmark = pytest.mark.gpu

N_EXP = 6


def work(eng, v, indices, modes, out, errors, barrier=None):
    """What one thread does with ITS context: whole exposures three ways, and thrower calls in between."""
    try:
        g = load_golden_psf("s128_t2")
        for n, i in enumerate(indices):
            pg = helpers.product_generator(v, i)
            desc = pg.build_descriptor(eng, rng_mode=modes[n % len(modes)], out_dtype=np.float32, **v.frame_kwargs(i))
            if barrier is not None:
                barrier.wait(timeout=60)              # both threads enter the library together, every round
            if n % 3 == 0:
                reads = eng.ctx.synthesize(desc)
            elif n % 3 == 1:
                eng.ctx.upload(2, desc)
                eng.ctx.run_front(2)
                eng.ctx.run_back(2)
                reads = eng.ctx.download(2)
            else:
                eng.ctx.upload(5, desc)
                eng.ctx.run(5)
                eng.ctx.fetch_async(5)
                reads = np.array(eng.ctx.wait(5))
            frame = eng.ctx.psf_apply(g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], g["nr"], g["nc"],
                                      g["test"] + i, g["threads"], rng_mode=_lib.RNG_REPLAY)
            out[i] = (reads.copy(), frame.copy())
    except BaseException as e:      # surfaced by the test
        errors.append(e)


def test_two_contexts_on_two_threads_equal_the_serial_frames():
    v = helpers.make_visit("small256", n_exposures=2 * N_EXP)
    modes = [_lib.RNG_SPLIT, _lib.RNG_PHILOX, _lib.RNG_REPLAY]
    mine = [list(range(0, 2 * N_EXP, 2)), list(range(1, 2 * N_EXP, 2))]
    # serial: one context, one thread
    serial, errors = {}, []
    eng0 = engine.Engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    try:
        for idx in mine:
            work(eng0, v, idx, modes, serial, errors)
    finally:
        eng0.close()
    assert not errors, errors
    # two contexts, two threads, at the same time
    engs = [engine.Engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY) for _ in range(2)]
    got, errors = {}, []
    barrier = threading.Barrier(2)
    try:
        threads = [threading.Thread(target=work, args=(engs[t], v, mine[t], modes, got, errors, barrier)) for t in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=600)
            assert not th.is_alive()
    finally:
        for e in engs:
            e.close()
    assert not errors, errors
    assert sorted(got) == sorted(serial) == list(range(2 * N_EXP))
    for i in serial:
        np.testing.assert_array_equal(got[i][0], serial[i][0], err_msg="exposure %d" % i)
        np.testing.assert_array_equal(got[i][1], serial[i][1], err_msg="thrower call %d" % i)
    assert np.abs(serial[0][0] - serial[1][0]).max() > 1.0          # (different exposures: the comparison means something)


def test_apply_psf_drop_in_from_several_python_threads():
    """`pyparallel.apply_psf` keeps ONE context per device for the whole process; the reference's Cython function holds
    the interpreter lock through its C call (pyparallel.pyx:14-38), the ctypes call does not -- the shim serialises its
    callers instead.  Four threads, sixteen calls each with their own `test` seed, started together: every frame equals the
    one a single thread gets."""
    from wayne_amd import pyparallel
    g = load_golden_psf("s128_t2")
    args = (g["counts"], g["x"], g["y"], g["ratio"], g["sl"], g["sh"], g["nr"], g["nc"])
    n_threads, n_calls = 4, 16
    serial = {s: pyparallel.apply_psf(*args, g["test"] + s, g["threads"]) for s in range(n_threads * n_calls)}
    got, errors = {}, []
    barrier = threading.Barrier(n_threads)

    def caller(t):
        try:
            barrier.wait(timeout=60)
            for s in range(t, n_threads * n_calls, n_threads):
                got[s] = pyparallel.apply_psf(*args, g["test"] + s, g["threads"])
        except BaseException as e:
            errors.append(e)

    threads = [threading.Thread(target=caller, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
        assert not th.is_alive()
    assert not errors, errors
    assert sorted(got) == sorted(serial)
    for s in serial:
        np.testing.assert_array_equal(got[s], serial[s], err_msg="call %d" % s)
    assert np.abs(serial[0] - serial[1]).max() > 0
    np.testing.assert_array_equal(serial[0], np.asarray(g["frame"], dtype=np.float64).ravel())   # and seed 0 is the reference's golden frame


def test_a_thread_editing_the_environment_does_not_change_a_live_context():
    # The knobs are frozen when the context is created (wayne_ctx_create reads WAYNE_* once; afterwards only
    # wayne_ctx_set_knob changes one): a thread that rewrites every WAYNE_* variable while another thread's context
    # uploads and runs exposures changes neither the frames NOR the launch shapes -- the knobs read back unchanged, and
    # knobs that would show in the frame's bookkeeping (lane_reach = 5 forces second runs) stay without effect.
    import os
    v = helpers.make_visit("small256", n_exposures=4)
    eng = engine.Engine(0, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ, v.SUBARRAY)
    names = {"WAYNE_BATCH": "7", "WAYNE_THIN": "1", "WAYNE_NO_ACC_BOX": "1", "WAYNE_LANE_REACH": "5", "WAYNE_THROW_WGS": "33",
             "WAYNE_TILE_INTS": "300", "WAYNE_KEEP_NARROW": "1", "WAYNE_NO_FUSE": "1", "WAYNE_STREAMS": "1"}
    saved = {k: os.environ.get(k) for k in names}
    try:
        def frames():
            out = []
            for i in range(4):
                pg = helpers.product_generator(v, i)
                desc = pg.build_descriptor(eng, out_dtype=np.float32, **v.frame_kwargs(i))
                eng.ctx.upload(i, desc)
                eng.ctx.run(i)
            eng.ctx.synchronize()
            for i in range(4):
                out.append(eng.ctx.download(i))
            return np.stack(out)
        want = frames()
        knobs_before = {k: eng.ctx.get_knob(k) for k in _lib.KNOBS}
        reruns_before = eng.ctx.reruns
        stop = threading.Event()
        # every variable is given a value ONCE here, while nothing of this process is inside the library: the meddler
        # below then only replaces values of names that exist (a pointer swap in libc's table), never adds or removes one
        # -- growing or shrinking the table under a concurrent reader elsewhere in the process (the HIP runtime reads its
        # own variables with getenv) would be a hazard of this TEST, not of the product
        for k, val in names.items():
            os.environ[k] = val

        def meddle():
            flip = 0
            while not stop.is_set():
                for k, val in names.items():
                    os.environ[k] = val if flip & 1 else "2"
                flip += 1
        t = threading.Thread(target=meddle)
        t.start()
        try:
            for _ in range(5):
                np.testing.assert_array_equal(frames(), want)
        finally:
            stop.set()
            t.join(60)
        assert {k: eng.ctx.get_knob(k) for k in _lib.KNOBS} == knobs_before
        assert eng.ctx.reruns == reruns_before                  # lane_reach = 5 never reached the context
        # ... while the explicit call does change the launch sequence (and still not the frames)
        eng.ctx.set_knob("lane_reach", 5)
        np.testing.assert_array_equal(frames(), want)
        assert eng.ctx.reruns > reruns_before
        # and a context created NOW does pick the variables up -- the one place they are read
        os.environ["WAYNE_LANE_REACH"] = "5"
        os.environ["WAYNE_BATCH"] = "7"
        c2 = _lib.Context(0)
        try:
            assert c2.get_knob("lane_reach") == 5 and c2.get_knob("batch") == 7
        finally:
            c2.close()
    finally:
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
        eng.close()
