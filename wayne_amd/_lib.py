"""ctypes binding of libwayne_hip.so (include/wayne_hip.h).

The library is the only compute path.  If it is missing it is built with
hipcc (wayne_amd/build.py); if that fails, or no gfx950 GPU is present when a
context is requested, this raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import build as _build

# status codes (wayne_hip.h)
OK, E_INVALID, E_NEGATIVE, E_OVERFLOW, E_NOMEM, E_HIP, E_NODEVICE, E_STATE = 0, -1, -2, -3, -4, -5, -6, -7
RNG_REPLAY, RNG_PHILOX, RNG_SPLIT = 0, 1, 2
F_ADD_FLAT = 1 << 0
F_ADD_GAIN_VARIATIONS = 1 << 1
F_ADD_NON_LINEAR = 1 << 2
F_CLIP_DET_LIMITS = 1 << 3
F_ADD_READ_NOISE = 1 << 4
F_ADD_STELLAR_NOISE = 1 << 5
F_ADD_DARK = 1 << 6
F_ADD_INITIAL_BIAS = 1 << 7
F_OUT_F64 = 1 << 16
F_EXACT_SAMPLERS = 1 << 17
PROF_KERNELS = 8
ABI_VERSION = 7


class WayneError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, "wayne_hip status %d: %s" % (status, msg))
        self.status = status


class WayneNoDeviceError(WayneError):
    pass


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class GrismDesc(C.Structure):
    _fields_ = [("trace_coeff", C.c_double * 9), ("wl_solution", C.c_double * 9),
                ("psf_ratio_poly", C.c_double * 4), ("psf_sigmal_poly", C.c_double * 4),
                ("psf_sigmah_poly", C.c_double * 4), ("n_sens", C.c_int),
                ("sens_wl_um", _dp), ("sens_val", _dp),
                ("flat_wmin", C.c_double), ("flat_wmax", C.c_double)]


class Calibration(C.Structure):
    _fields_ = [("subarray", C.c_int), ("n_reads", C.c_int), ("flat", _fp * 4), ("pfl", _fp),
                ("sky", _fp), ("lin", _fp * 4), ("dark_sci", _fp), ("dark_err", _fp),
                ("zero_read", _dp)]


class ExposureDesc(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("exposure_index", C.c_uint32), ("rng_mode", C.c_int),
                ("threads_compat", C.c_int), ("flags", C.c_uint32), ("sub_scale", C.c_int),
                ("n_wl", C.c_int), ("wl_um", _dp), ("flux", _dp), ("depth", _dp),
                ("n_samples", C.c_int), ("x_ref", _dp), ("y_ref", _dp), ("dur_ms", _dp),
                ("replay_seed", _ip), ("sample_read", _ip),
                ("n_reads", C.c_int), ("read_dt_s", _dp),
                ("sky_ct_s", C.c_double), ("cosmic_rate", C.c_double), ("scale_factor", C.c_double),
                ("noise_mean", C.c_double), ("noise_std", C.c_double),
                ("thrower_margin", C.c_int), ("thrower_splits", C.c_int),
                ("lc_z", _dp), ("lc_hidden", _dp), ("lc_rp", _dp), ("lc_ld", C.c_double * 4)]


class Profile(C.Structure):
    _fields_ = [("name", C.c_char_p * PROF_KERNELS), ("launches", C.c_uint64 * PROF_KERNELS),
                ("ms", C.c_double * PROF_KERNELS), ("electrons", C.c_uint64)]


# every symbol include/wayne_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "wayne_abi_version": (C.c_int, []),
    "wayne_strerror": (C.c_char_p, [C.c_int]),
    "wayne_build_flags": (C.c_char_p, []),
    "wayne_device_count": (C.c_int, []),
    "wayne_ctx_create": (_vp, [C.c_int, C.POINTER(C.c_int)]),
    "wayne_ctx_destroy": (None, [_vp]),
    "wayne_last_error": (C.c_char_p, [_vp]),
    "wayne_ctx_synchronize": (C.c_int, [_vp]),
    "wayne_ctx_stream": (_vp, [_vp]),
    "wayne_ctx_set_knob": (C.c_int, [_vp, C.c_char_p, C.c_longlong]),
    "wayne_ctx_get_knob": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_longlong)]),
    "wayne_psf_apply": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                  C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "wayne_psf_apply_ex": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                     C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "wayne_ctx_set_grism": (C.c_int, [_vp, C.POINTER(GrismDesc)]),
    "wayne_ctx_set_calibration": (C.c_int, [_vp, C.POINTER(Calibration)]),
    "wayne_ctx_slots": (C.c_int, [_vp]),
    "wayne_exposure_upload": (C.c_int, [_vp, C.c_int, C.POINTER(ExposureDesc)]),
    "wayne_exposure_run": (C.c_int, [_vp, C.c_int]),
    "wayne_exposure_run_checked": (C.c_int, [_vp, C.c_int]),
    "wayne_exposure_status": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "wayne_ctx_reruns": (C.c_ulonglong, [_vp]),
    "wayne_exposure_download": (C.c_int, [_vp, C.c_int, _vp]),
    "wayne_exposure_fetch_async": (C.c_int, [_vp, C.c_int]),
    "wayne_exposure_wait": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p)]),
    "wayne_exposure_device_reads": (_vp, [_vp, C.c_int]),
    "wayne_exposure_synthesize": (C.c_int, [_vp, C.POINTER(ExposureDesc), _vp]),
    "wayne_exposure_debug_fetch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "wayne_exposure_debug_depth": (C.c_int, [_vp, C.c_int, _vp]),
    "wayne_exposure_debug_boxes": (C.c_int, [_vp, C.c_int, _vp, _vp, C.POINTER(C.c_int)]),
    "wayne_exposure_run_front": (C.c_int, [_vp, C.c_int]),
    "wayne_exposure_run_back": (C.c_int, [_vp, C.c_int]),
    "wayne_exposure_ramp_variant": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int]),
    "wayne_profile_enable": (C.c_int, [_vp, C.c_int]),
    "wayne_profile_select": (C.c_int, [_vp, C.c_uint]),
    "wayne_profile_reset": (C.c_int, [_vp]),
    "wayne_profile_get": (C.c_int, [_vp, C.POINTER(Profile)]),
    "wayne_philox4x32": (None, [_vp, _vp, _vp]),
    "wayne_host_sample_draws": (None, [C.c_uint32, C.c_uint32, C.c_int, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load (building first if needed) libwayne_hip.so.  Raises on failure."""
    global _lib, _lib_path
    if _lib is None:
        path = _build.LIB
        override = os.environ.get("WAYNE_HIP_LIB")
        if override:
            # a library built elsewhere with other flags (A/B builds, the negative-control build of
            # tests/test_extremes_gpu.py): loaded as it is, never rebuilt; a missing file is an OSError
            path = override
        elif _build.stale():
            if os.path.exists(_build.HIPCC):
                _build.build(verbose=False)
            elif not os.path.exists(path):
                raise ImportError("libwayne_hip.so is not built and hipcc is unavailable (%s)" % _build.HIPCC)
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the ABI lost a symbol
            f.restype = res
            f.argtypes = args
        if L.wayne_abi_version() != ABI_VERSION:
            raise ImportError("libwayne_hip.so ABI version mismatch")
        flags = L.wayne_build_flags().decode().split()
        if flags and os.environ.get("WAYNE_ALLOW_FLAGGED_LIB", "") != "1":
            # a negative-control or timing build (its sources say "wrong frames"): never by accident
            raise ImportError("%s was built with %s: not the product's library (tests and measurement scripts that mean "
                              "to load it set WAYNE_ALLOW_FLAGGED_LIB=1)" % (path, " ".join(flags)))
        _lib = L
        _lib_path = path
    return _lib


_lib_path = None


def library_info():
    """(path, build flags) of the loaded library: what a measurement or a written file names as its producer."""
    L = load()
    return _lib_path or _build.LIB, L.wayne_build_flags().decode().strip()


def ptr(a, ctype=None):
    """Raw pointer of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    if ctype is None:
        return a.ctypes.data_as(C.c_void_p)
    return a.ctypes.data_as(C.POINTER(ctype))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


KNOBS = ("tile_ints", "batch", "thin", "no_acc_box", "lane_reach", "throw_wgs", "keep_narrow", "no_fuse", "fork_narrow",
         "streams", "upload_timing", "ramp_reads")
_live_contexts = weakref.WeakSet()
_knob_defaults = {}


def set_knob_all(name, value):
    """Set a tuning / test knob on every live context of this process and on those created later (None: back to the
    library's own choice).  The explicit, per-process replacement of editing WAYNE_* environment variables under a
    running context -- the library reads those once, in wayne_ctx_create."""
    if name not in KNOBS:
        raise KeyError("no knob named %r" % (name,))
    if value is None:
        _knob_defaults.pop(name, None)
    else:
        _knob_defaults[name] = int(value)
    for c in list(_live_contexts):
        if getattr(c, "_h", None):
            c.set_knob(name, value)


def reset_knobs_all():
    for name in list(_knob_defaults):
        set_knob_all(name, None)


class Context(object):
    """One wayne_ctx: a GPU, a HIP stream and the HBM buffers of the path."""

    def __init__(self, device=0):
        self._L = load()
        st = C.c_int(0)
        self._h = self._L.wayne_ctx_create(int(device), C.byref(st))
        if not self._h:
            msg = self._L.wayne_strerror(st.value).decode()
            cls = WayneNoDeviceError if st.value == E_NODEVICE else WayneError
            raise cls(st.value, "wayne_ctx_create(device=%d): %s -- the HIP path is the only path, "
                                "there is no CPU fallback" % (device, msg))
        self.device = device
        self._keep = []  # arrays referenced by descriptors during a call
        _live_contexts.add(self)
        for name, value in _knob_defaults.items():
            self.set_knob(name, value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.wayne_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, status):
        if status != OK:
            raise WayneError(status, self._L.wayne_last_error(self._h).decode())

    def synchronize(self):
        """Wait for everything enqueued and settle it: afterwards every slot's reads are complete (an exposure that
        met a bin beyond its launch sequence's reach has been run a second time; see wayne_ctx_synchronize)."""
        self.check(self._L.wayne_ctx_synchronize(self._h))

    def set_knob(self, name, value):
        """Tuning / test knob of THIS context (value None or < 0: the library's own choice).  The environment
        (WAYNE_<NAME>) is read once, at creation."""
        self.check(self._L.wayne_ctx_set_knob(self._h, name.encode(), -1 if value is None else int(value)))

    def get_knob(self, name):
        v = C.c_longlong(0)
        self.check(self._L.wayne_ctx_get_knob(self._h, name.encode(), C.byref(v)))
        return None if v.value < 0 else int(v.value)

    @property
    def stream(self):
        return self._L.wayne_ctx_stream(self._h)

    # -- inner boundary -----------------------------------------------------
    def psf_apply(self, counts, x, y, ratio, sl, sh, nr, nc, seed, threads=1, rng_mode=RNG_REPLAY,
                  exposure=0, subsample=0, exact_samplers=False):
        counts = i32(counts)
        x, y, ratio, sl, sh = f64(x), f64(y), f64(ratio), f64(sl), f64(sh)
        n = counts.size
        if not all(a.size == n for a in (x, y, ratio, sl, sh)):
            raise ValueError("apply_psf: arrays differ in length")
        out = np.empty(max(int(nr) * int(nc), 0), dtype=np.int32)
        self.check(self._L.wayne_psf_apply_ex(self._h, ptr(counts), n, ptr(x), ptr(y), ptr(ratio), ptr(sl),
                                              ptr(sh), int(nr), int(nc), int(seed) & 0xFFFFFFFF, int(threads),
                                              int(rng_mode), int(exposure), int(subsample),
                                              F_EXACT_SAMPLERS if exact_samplers else 0, ptr(out)))
        return out

    # -- grism / calibration --------------------------------------------------
    def set_grism(self, trace_coeff, wl_solution, psf_ratio_poly, psf_sigmal_poly, psf_sigmah_poly,
                  sens_wl_um, sens_val, flat_wmin, flat_wmax):
        g = GrismDesc()
        g.trace_coeff[:] = list(map(float, trace_coeff))
        g.wl_solution[:] = list(map(float, wl_solution))
        g.psf_ratio_poly[:] = list(map(float, psf_ratio_poly))
        g.psf_sigmal_poly[:] = list(map(float, psf_sigmal_poly))
        g.psf_sigmah_poly[:] = list(map(float, psf_sigmah_poly))
        wl, val = f64(sens_wl_um), f64(sens_val)
        if wl.size != val.size:
            raise ValueError("sensitivity table: wl and val differ in length")
        g.n_sens = wl.size
        g.sens_wl_um = ptr(wl, C.c_double)
        g.sens_val = ptr(val, C.c_double)
        g.flat_wmin, g.flat_wmax = float(flat_wmin), float(flat_wmax)
        self.check(self._L.wayne_ctx_set_grism(self._h, C.byref(g)))

    def set_calibration(self, subarray, n_reads, flat=None, pfl=None, sky=None, lin=None, dark_sci=None,
                        dark_err=None, zero_read=None):
        N = 1014 if subarray == 1024 else subarray
        S = N + 10
        k = Calibration()
        k.subarray, k.n_reads = int(subarray), int(n_reads)
        keep = []

        def plane(a, shape, name, dt=np.float32):
            a = np.ascontiguousarray(a, dtype=dt)
            if a.shape != shape:
                raise ValueError("%s: expected shape %s, got %s" % (name, shape, a.shape))
            keep.append(a)
            return a

        if flat is not None:
            for i in range(4):
                k.flat[i] = ptr(plane(flat[i], (N, N), "flat[%d]" % i), C.c_float)
        if pfl is not None:
            k.pfl = ptr(plane(pfl, (N, N), "pfl"), C.c_float)
        if sky is not None:
            k.sky = ptr(plane(sky, (N, N), "sky"), C.c_float)
        if lin is not None:
            for i in range(4):
                k.lin[i] = ptr(plane(lin[i], (S, S), "lin[%d]" % i), C.c_float)
        if dark_sci is not None and dark_err is not None:
            k.dark_sci = ptr(plane(dark_sci, (n_reads, S, S), "dark_sci"), C.c_float)
            k.dark_err = ptr(plane(dark_err, (n_reads, S, S), "dark_err"), C.c_float)
        if zero_read is not None:
            k.zero_read = ptr(plane(zero_read, (S, S), "zero_read", np.float64), C.c_double)
        self.check(self._L.wayne_ctx_set_calibration(self._h, C.byref(k)))
        self.N, self.S, self.R = N, S, int(n_reads)

    # -- exposures -------------------------------------------------------------
    def make_desc(self, *a, **k):
        return make_desc(*a, **k)

    def upload(self, slot, desc):
        self.check(self._L.wayne_exposure_upload(self._h, int(slot), C.byref(desc)))
        self._slot_meta = getattr(self, "_slot_meta", {})
        self._slot_meta[slot] = (desc.n_samples, desc.n_wl, desc.n_reads, bool(desc.flags & F_OUT_F64))

    def run(self, slot):
        self.check(self._L.wayne_exposure_run(self._h, int(slot)))

    def run_checked(self, slot):
        """run + wait for the slot + settle it: its reads in HBM are complete on return."""
        self.check(self._L.wayne_exposure_run_checked(self._h, int(slot)))

    def run_front(self, slot):
        self.check(self._L.wayne_exposure_run_front(self._h, int(slot)))

    def run_back(self, slot):
        self.check(self._L.wayne_exposure_run_back(self._h, int(slot)))

    def ramp_variant(self, slot):
        """Name of the k_ramp instantiation the slot's back half launches, as a kernel trace prints it."""
        buf = C.create_string_buffer(128)
        self.check(self._L.wayne_exposure_ramp_variant(self._h, int(slot), buf, len(buf)))
        return buf.value.decode()

    def status(self, slot):
        """Status word of the slot's last run (0 = complete; see wayne_exposure_status).  Synchronises."""
        st = C.c_int(0)
        self.check(self._L.wayne_exposure_status(self._h, int(slot), C.byref(st)))
        return st.value

    @property
    def reruns(self):
        """Exposures this context ran a second time (a bin beyond the first launch sequence's reach)."""
        return int(self._L.wayne_ctx_reruns(self._h))

    def download(self, slot):
        K, W, R, f64out = self._slot_meta[slot]
        out = np.empty((R + 1, self.S, self.S), dtype=np.float64 if f64out else np.float32)
        self.check(self._L.wayne_exposure_download(self._h, int(slot), ptr(out)))
        return out

    def fetch_async(self, slot):
        """Enqueue the copy of the slot's reads into its pinned host buffer (returns at once)."""
        self.check(self._L.wayne_exposure_fetch_async(self._h, int(slot)))

    def wait(self, slot):
        """Block until the slot's work is done -> its reads as a numpy VIEW of the pinned buffer
        (valid until the slot is uploaded again; copy it to keep it)."""
        K, W, R, f64out = self._slot_meta[slot]
        p = C.c_void_p()
        self.check(self._L.wayne_exposure_wait(self._h, int(slot), C.byref(p)))
        ct = C.c_double if f64out else C.c_float
        n = (R + 1) * self.S * self.S
        buf = (ct * n).from_address(p.value)
        return np.frombuffer(buf, dtype=np.float64 if f64out else np.float32).reshape(R + 1, self.S, self.S)

    def synthesize(self, desc):
        self.upload(0, desc)
        self.run(0)
        return self.download(0)

    def debug_depth(self, slot):
        K, W, R, _ = self._slot_meta[slot]
        out = np.empty((K, W), dtype=np.float64)
        self.check(self._L.wayne_exposure_debug_depth(self._h, int(slot), ptr(out)))
        return out

    def debug_boxes(self, slot):
        """(use_box, boxes[16, 4], segments[16]): which accumulators k_ramp loads for the slot's exposure."""
        boxes = np.zeros((16, 4), dtype=np.int32)
        seg = np.zeros(16, dtype=np.int32)
        use = C.c_int(0)
        self.check(self._L.wayne_exposure_debug_boxes(self._h, int(slot), ptr(boxes), ptr(seg), C.byref(use)))
        return bool(use.value), boxes, seg

    def debug_fetch(self, slot, acc=False):
        K, W, R, _ = self._slot_meta[slot]
        counts = np.empty((K, W), dtype=np.int32)
        x = np.empty((K, W), dtype=np.float64)
        y = np.empty((K, W), dtype=np.float64)
        a = np.empty((R, self.S, self.S), dtype=np.float64) if acc else None
        self.check(self._L.wayne_exposure_debug_fetch(self._h, int(slot), ptr(counts), ptr(x), ptr(y), ptr(a)))
        return counts, x, y, a

    # -- measurement -----------------------------------------------------------
    def profile_enable(self, on=True):
        self.check(self._L.wayne_profile_enable(self._h, 1 if on else 0))

    def profile_select(self, names=None):
        """Time only the named kernels (e.g. ["k_ramp"]); None = all."""
        if names is None:
            mask = 0xFFFFFFFF
        else:
            order = list(self.profile_get().keys())
            mask = 0
            for n in names:
                mask |= 1 << order.index(n)
        self.check(self._L.wayne_profile_select(self._h, mask))

    def profile_reset(self):
        self.check(self._L.wayne_profile_reset(self._h))

    def profile_get(self):
        p = Profile()
        self.check(self._L.wayne_profile_get(self._h, C.byref(p)))
        out = {}
        for i in range(PROF_KERNELS):
            out[p.name[i].decode()] = {"launches": int(p.launches[i]), "ms": float(p.ms[i])}
        out["electrons"] = int(p.electrons)
        return out


def make_desc(seed, exposure_index, flags, sub_scale, wl_um, flux, depth, x_ref, y_ref, dur_ms,
              sample_read, read_dt_s, replay_seed=None, rng_mode=RNG_PHILOX, threads_compat=1,
              sky_ct_s=0.0, cosmic_rate=-1.0, scale_factor=1.0, noise_mean=0.0, noise_std=0.0,
              thrower_margin=0, thrower_splits=0, lc_z=None, lc_hidden=None, lc_rp=None, lc_ld=None):
    d = ExposureDesc()
    keep = []

    def arr(a, conv):
        a = conv(a)
        keep.append(a)
        return a

    wl_um, flux = arr(wl_um, f64), arr(flux, f64)
    x_ref, y_ref, dur_ms = arr(x_ref, f64), arr(y_ref, f64), arr(dur_ms, f64)
    sample_read, read_dt_s = arr(sample_read, i32), arr(read_dt_s, f64)
    W, K = wl_um.size, x_ref.size
    if flux.size != W or y_ref.size != K or dur_ms.size != K or sample_read.size != K:
        raise ValueError("exposure descriptor: inconsistent array lengths")
    d.seed, d.exposure_index = int(seed) & 0xFFFFFFFF, int(exposure_index) & 0xFFFFFFFF
    d.rng_mode, d.threads_compat = int(rng_mode), int(threads_compat)
    d.flags, d.sub_scale = int(flags), int(sub_scale)
    d.n_wl, d.wl_um, d.flux = W, ptr(wl_um, C.c_double), ptr(flux, C.c_double)
    if depth is not None:
        depth = arr(depth, f64)
        if depth.shape != (K, W):
            raise ValueError("depth must have shape (K, W)")
        d.depth = ptr(depth, C.c_double)
    d.n_samples = K
    d.x_ref, d.y_ref, d.dur_ms = ptr(x_ref, C.c_double), ptr(y_ref, C.c_double), ptr(dur_ms, C.c_double)
    if replay_seed is not None:
        replay_seed = arr(replay_seed, i32)
        d.replay_seed = ptr(replay_seed, C.c_int32)
    d.sample_read = ptr(sample_read, C.c_int32)
    d.n_reads, d.read_dt_s = read_dt_s.size, ptr(read_dt_s, C.c_double)
    d.sky_ct_s, d.cosmic_rate = float(sky_ct_s), float(cosmic_rate)
    d.scale_factor, d.noise_mean, d.noise_std = float(scale_factor), float(noise_mean), float(noise_std)
    d.thrower_margin, d.thrower_splits = int(thrower_margin), int(thrower_splits)
    if lc_z is not None:
        if depth is not None:
            raise ValueError("give either depth or lc_z")
        lc_z, lc_rp = arr(lc_z, f64), arr(lc_rp, f64)
        if lc_z.size != K or lc_rp.size != W:
            raise ValueError("lc_z must have K and lc_rp W elements")
        d.lc_z, d.lc_rp = ptr(lc_z, C.c_double), ptr(lc_rp, C.c_double)
        if lc_hidden is not None:
            lc_hidden = arr(lc_hidden, f64)
            d.lc_hidden = ptr(lc_hidden, C.c_double)
        d.lc_ld[:] = [float(v) for v in lc_ld]
    d._keep = keep
    return d


def device_count():
    return load().wayne_device_count()


def philox4x32(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    load().wayne_philox4x32(ptr(ctr), ptr(key), ptr(out))
    return out


def host_sample_draws(seed, exposure, n_samples):
    """(z_x, z_y, rand_seed) of the Philox HOST stage for one exposure."""
    zx = np.empty(n_samples, dtype=np.float64)
    zy = np.empty(n_samples, dtype=np.float64)
    rs = np.empty(n_samples, dtype=np.int32)
    load().wayne_host_sample_draws(int(seed) & 0xFFFFFFFF, int(exposure) & 0xFFFFFFFF, int(n_samples),
                                   ptr(zx), ptr(zy), ptr(rs))
    return zx, zy, rs


_default_ctx = {}
_default_ctx_lock = threading.Lock()


def default_context(device=0):
    """Process-wide context per device (created on first use; creation is serialised)."""
    with _default_ctx_lock:
        if device not in _default_ctx:
            ctx = Context(device)
            ctx.call_lock = threading.Lock()   # for callers that share it between threads (pyparallel.apply_psf)
            _default_ctx[device] = ctx
        return _default_ctx[device]
