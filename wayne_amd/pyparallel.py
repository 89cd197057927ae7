"""Drop-in for the reference's Cython module ``wayne.pyparallel``.

``apply_psf`` has the reference's signature and return type
(wayne/pyparallel.pyx:14-38): float64 array of length NR*NC holding the
electron-count frame, to be reshaped to (NR, NC) by the caller
(exposure_generator.py:636-639).  The work happens in the HIP thrower kernel
through ``wayne_psf_apply`` (include/wayne_hip.h).

With the defaults the frame is bit-identical to the reference's for the same
``(test, threads)``: rng_mode 0 replays glibc ``rand_r`` and the OpenMP
partition of pyparallel_menu.c:40-64 on the GPU.
"""
import numpy as np

from . import _lib

RNG_REPLAY = _lib.RNG_REPLAY
RNG_PHILOX = _lib.RNG_PHILOX


def apply_psf(counts, pos_x, pos_y, ratio_psf, sigmal_psf, sigmah_psf, NR, NC, test, threads,
              rng_mode=RNG_REPLAY, device=0, exposure=0, subsample=0):
    # the shim converts every count with C's (int) cast (pyparallel.pyx:23-25)
    counts = np.asarray(counts)
    if counts.dtype.kind == "f":
        counts = np.trunc(counts)
    # One process-wide context per device, and calls on one context are not thread-safe (include/wayne_hip.h): the
    # reference's Cython function holds the interpreter lock for the whole call, the ctypes call releases it, so callers
    # on several Python threads are serialised here instead.
    ctx = _lib.default_context(device)
    with ctx.call_lock:
        frame = ctx.psf_apply(counts.astype(np.int32), pos_x, pos_y, ratio_psf, sigmal_psf, sigmah_psf,
                              NR, NC, test, threads, rng_mode, exposure, subsample)
    return frame.astype(np.float64)  # pyparallel.pyx:31-34
