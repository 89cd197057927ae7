"""ctypes access to the oracle's C libraries.  TEST INFRASTRUCTURE ONLY.

  libwayne_oracle.so        this repo's C restatement (oracle/psf_oracle.c, noise_oracle.c, split_oracle.c,
                            lc_oracle.c)
  _ref/libwayne_ref_psf.so  the reference's own wayne/pyparallel_menu.c compiled
                            unmodified by oracle/Makefile (present only where
                            /root/reference was available at build time, or
                            shipped prebuilt to the GPU box)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.join(HERE, "libwayne_oracle.so")
_REF = os.path.join(HERE, "_ref", "libwayne_ref_psf.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """(Re)build the oracle libraries with oracle/Makefile."""
    srcs = [os.path.join(HERE, f) for f in ("psf_oracle.c", "noise_oracle.c", "split_oracle.c", "lc_oracle.c",
                                             "Makefile")]
    stale = force or not os.path.exists(_ORACLE) or any(
        os.path.getmtime(s) > os.path.getmtime(_ORACLE) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "-B", "liboracle"])
    if os.path.exists("/root/reference/wayne/pyparallel_menu.c") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_ORACLE)
        L.wayne_oracle_psf.restype = C.c_int
        L.wayne_oracle_psf.argtypes = [_i32p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
        L.wayne_oracle_psf_philox.restype = C.c_int
        L.wayne_oracle_psf_philox.argtypes = [_i32p, C.c_int, _f64p, _f64p, _f64p, _f32p, _f32p,
                                              C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _i32p]
        L.wayne_oracle_philox4x32.restype = None
        L.wayne_oracle_philox4x32.argtypes = [_u32p, _u32p, _u32p]
        L.wayne_oracle_rand_r.restype = C.c_int
        L.wayne_oracle_rand_r.argtypes = [C.POINTER(C.c_uint32)]
        L.wayne_oracle_poisson_f64.restype = None
        L.wayne_oracle_poisson_f64.argtypes = [_f64p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_uint32, _f64p]
        L.wayne_oracle_poisson_counts_f64.restype = None
        L.wayne_oracle_poisson_counts_f64.argtypes = [_f64p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      C.c_uint32, C.c_uint32, _f64p]
        L.wayne_oracle_seed_streams.restype = None
        L.wayne_oracle_seed_streams.argtypes = [_u32p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, _u32p]
        L.wayne_oracle_poisson_sky_step.restype = None
        L.wayne_oracle_poisson_sky_step.argtypes = [_f32p, C.c_int64, _u32p, _f64p]
        L.wayne_oracle_normal_step.restype = None
        L.wayne_oracle_normal_step.argtypes = [C.c_int64, _u32p, _f32p, _f32p]
        L.wayne_oracle_xo_next.restype = C.c_uint32
        L.wayne_oracle_xo_next.argtypes = [_u32p]
        L.wayne_oracle_philox_blocks.restype = None
        L.wayne_oracle_philox_blocks.argtypes = [_u32p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_uint32, _u32p]
        L.wayne_oracle_sky_alias_table.restype = None
        L.wayne_oracle_sky_alias_table.argtypes = [C.c_double, _u32p]
        L.wayne_oracle_sky_alias_step.restype = None
        L.wayne_oracle_sky_alias_step.argtypes = [_f32p, _f32p, _i32p, _u32p, C.c_int64, _u32p, _f64p]
        L.wayne_oracle_psf_split.restype = C.c_int
        L.wayne_oracle_psf_split.argtypes = [_i32p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_int, C.c_int,
                                             C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _i32p]
        L.wayne_oracle_upper_tail.restype = C.c_float
        L.wayne_oracle_upper_tail.argtypes = [C.c_float]
        L.wayne_oracle_binomial_vec.restype = None
        L.wayne_oracle_binomial_vec.argtypes = [_f32p, _f32p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, _f32p]
        L.wayne_oracle_binomial_trace.restype = None
        L.wayne_oracle_binomial_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.wayne_oracle_binomial_trace_len.restype = C.c_int
        L.wayne_oracle_binomial_trace_len.argtypes = []
        L.wayne_oracle_xo_pairs.restype = None
        L.wayne_oracle_xo_pairs.argtypes = [_u32p, C.c_int64, _u32p]
        L.wayne_oracle_lc_deficit.restype = None
        L.wayne_oracle_lc_deficit.argtypes = [C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_int, _f64p]
        L.wayne_oracle_lc_hidden.restype = None
        L.wayne_oracle_lc_hidden.argtypes = [C.c_int, _f64p, _f64p, _f64p]
        L.wayne_oracle_lc_depths.restype = None
        L.wayne_oracle_lc_depths.argtypes = [C.c_int, C.c_int, _f64p, C.c_void_p, _f64p, _f64p, C.c_int, _f64p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    """The reference's compiled PSF(); None when oracle/_ref was not built."""
    global _ref
    if _ref is None and have_ref():
        L = C.CDLL(_REF)
        L.PSF.restype = C.POINTER(C.c_int)
        L.PSF.argtypes = [_i32p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p,
                          C.c_int, C.c_int, C.c_int, C.c_int]
        _ref = L
    return _ref


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.free.restype = None


def _prep(counts, x, y, ratio, sl, sh):
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, ratio, sl, sh)]
    n = counts.size
    assert all(a.size == n for a in arrs)
    return (counts,) + tuple(arrs)


def psf_oracle(counts, x, y, ratio, sl, sh, nr, nc, test, threads):
    """The repo's restatement of PSF() -> int32[nr*nc] (raises on invalid input)."""
    counts, x, y, ratio, sl, sh = _prep(counts, x, y, ratio, sl, sh)
    out = np.empty(nr * nc, dtype=np.int32)
    rc = lib().wayne_oracle_psf(counts, counts.size, x, y, ratio, sl, sh, nr, nc, int(test), int(threads), out)
    if rc != 0:
        raise ValueError("wayne_oracle_psf: status %d" % rc)
    return out


def psf_reference(counts, x, y, ratio, sl, sh, nr, nc, test, threads):
    """The reference's own compiled PSF() (wayne/pyparallel_menu.c) -> int32[nr*nc]."""
    L = ref()
    if L is None:
        raise RuntimeError("oracle/_ref/libwayne_ref_psf.so not built")
    counts, x, y, ratio, sl, sh = _prep(counts, x, y, ratio, sl, sh)
    p = L.PSF(counts, counts.size, x, y, ratio, sl, sh, nr, nc, int(test), int(threads))
    out = np.ctypeslib.as_array(p, shape=(nr * nc,)).astype(np.int32, copy=True)
    _libc.free(C.cast(p, C.c_void_p))
    return out


def psf_philox_oracle(counts, x, y, ratio, sl, sh, nr, nc, seed, exposure, subsample):
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    sl32, sh32 = (np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32)) for a in (sl, sh))
    x64, y64 = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y))   # split into pixel + fraction in the C
    ratio = np.ascontiguousarray(ratio, dtype=np.float64)
    out = np.empty(nr * nc, dtype=np.int32)
    rc = lib().wayne_oracle_psf_philox(counts, counts.size, x64, y64, ratio, sl32, sh32, nr, nc,
                                       int(seed), int(exposure), int(subsample), out)
    if rc != 0:
        raise ValueError("wayne_oracle_psf_philox: status %d" % rc)
    return out


def psf_split_oracle(counts, x, y, ratio, sl, sh, n, seed, exposure, subsample, split_min=32, lane_max=4096):
    """The thrower's default mode (WAYNE_RNG_SPLIT) on the CPU, same counters as the device."""
    counts, x, y, ratio, sl, sh = _prep(counts, x, y, ratio, sl, sh)
    out = np.empty(n * n, dtype=np.int32)
    rc = lib().wayne_oracle_psf_split(counts, counts.size, x, y, ratio, sl, sh, n, int(split_min), int(lane_max),
                                      int(seed), int(exposure), int(subsample), out)
    if rc != 0:
        raise ValueError("wayne_oracle_psf_split: status %d" % rc)
    return out


def psf_split_trace(counts, x, y, ratio, sl, sh, n, seed, exposure, subsample, perturb_at=-1, perturb_rel=0.0,
                    cap=1 << 20):
    """psf_split_oracle with every binomial call of the run recorded: -> (frame, calls[:, (n, p, result)]).
    Call number `perturb_at` has its probability multiplied by (1 + perturb_rel)."""
    buf = np.zeros((cap, 3), dtype=np.float32)
    L = lib()
    L.wayne_oracle_binomial_trace(buf.ctypes.data_as(C.c_void_p), cap, int(perturb_at), float(perturb_rel))
    try:
        frame = psf_split_oracle(counts, x, y, ratio, sl, sh, n, seed, exposure, subsample)
        used = L.wayne_oracle_binomial_trace_len()
    finally:
        L.wayne_oracle_binomial_trace(None, 0, -1, 0.0)
    return frame, buf[:min(used, cap)].copy()


def binomial_vec(n, p, seed, subsample=0, exposure=0):
    """Binomial(n[i], p[i]) from the STAGE_NARROW stream of element i (fp32 sampler)."""
    n = np.ascontiguousarray(n, dtype=np.float32)
    p = np.ascontiguousarray(p, dtype=np.float32)
    out = np.empty(n.size, dtype=np.float32)
    lib().wayne_oracle_binomial_vec(n, p, n.size, int(seed), int(subsample), int(exposure), out)
    return out


def upper_tail(t):
    """P(Z > t) of the standard normal as the split thrower computes it (float32 Chebyshev fit of erfc)."""
    return np.array([lib().wayne_oracle_upper_tail(float(v)) for v in np.atleast_1d(t)], dtype=np.float64)


def philox4x32(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().wayne_oracle_philox4x32(ctr, key, out)
    return out


def lc_deficit(z, p, ld, nodes=65):
    """1 - transit flux, shape (len(z), len(p)): 2-D integration over the planet's disk (oracle/lc_oracle.c)."""
    z = np.ascontiguousarray(z, dtype=np.float64).ravel()
    p = np.ascontiguousarray(p, dtype=np.float64).ravel()
    ld = np.ascontiguousarray(ld, dtype=np.float64)
    assert ld.size == 4
    out = np.empty((z.size, p.size))
    lib().wayne_oracle_lc_deficit(z.size, p.size, z, p, ld, int(nodes), out)
    return out


def lc_hidden(z, p):
    """Fraction of the planet's disk (radius p, separation z) inside the unit disk."""
    z, p = np.broadcast_arrays(np.asarray(z, dtype=np.float64), np.asarray(p, dtype=np.float64))
    zz, pp = np.ascontiguousarray(z).ravel(), np.ascontiguousarray(p).ravel()
    out = np.empty(zz.size)
    lib().wayne_oracle_lc_hidden(zz.size, zz, pp, out)
    return out.reshape(z.shape)


def lc_depths(z_tr, hidden, planet_spectrum, ld, nodes=65):
    """planet_depths (K, W) of observation.py:349-355, 442-443 from the oracle's own model."""
    z_tr = np.ascontiguousarray(z_tr, dtype=np.float64).ravel()
    spec = np.ascontiguousarray(planet_spectrum, dtype=np.float64).ravel()
    ld = np.ascontiguousarray(ld, dtype=np.float64)
    hp = None
    if hidden is not None:
        hidden = np.ascontiguousarray(hidden, dtype=np.float64).ravel()
        assert hidden.size == z_tr.size
        hp = hidden.ctypes.data_as(C.c_void_p)
    out = np.empty((z_tr.size, spec.size))
    lib().wayne_oracle_lc_depths(z_tr.size, spec.size, z_tr, hp, spec, ld, int(nodes), out)
    return out
