"""Driver of tests/native/plan_harness.cpp: the product's host-side launch planner (wayne_amd/csrc/host_plan.h, the
file wayne_hip.hip includes) compiled with g++ under AddressSanitizer + UndefinedBehaviorSanitizer and run on the CPU.

A `Batch` collects operations, `run()` writes them to a file, runs the harness ONCE and decodes one result per
operation.  A sanitizer report (or any crash) is a non-zero exit: run() raises with the harness's stderr.
"""
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE = os.path.join(HERE, "native")
BUILD = os.path.join(NATIVE, "_build")
VARIANTS = {"": "plan_harness", "ends": "plan_harness_negctl_ends", "stale": "plan_harness_negctl_stale",
            "nan": "plan_harness_negctl_nan"}
K_MAX_CHUNKS, K_SKY_ALIAS = 128, 256


def build():
    subprocess.run(["make", "-C", NATIVE, "-s", "all"], check=True, capture_output=True, text=True)


def binary(variant=""):
    return os.path.join(BUILD, VARIANTS[variant])


class HarnessError(RuntimeError):
    pass


def _f64(a):
    return np.ascontiguousarray(a, dtype="<f8").tobytes()


class Batch(object):
    def __init__(self):
        self.ops, self.kinds = [], []

    def set_grism(self, trace, wlsol, p_ratio, p_sigl, p_sigh, sens_wl=(), sens_val=()):
        sens_wl, sens_val = np.asarray(sens_wl, dtype=float), np.asarray(sens_val, dtype=float)
        assert sens_wl.size == sens_val.size
        self.ops.append(struct.pack("<i", 1) + _f64(trace) + _f64(wlsol) + _f64(p_ratio) + _f64(p_sigl) + _f64(p_sigh) +
                        struct.pack("<i", sens_wl.size) + _f64(sens_wl) + _f64(sens_val))
        self.kinds.append(("grism",))
        return len(self.ops) - 1

    def plan(self, S, sub_scale, rng_mode, wl, flux, x_ref, y_ref, dur_ms, sample_read, R, scale_factor=1.0):
        wl, flux = np.asarray(wl, dtype=float), np.asarray(flux, dtype=float)
        x_ref, y_ref, dur_ms = (np.asarray(a, dtype=float) for a in (x_ref, y_ref, dur_ms))
        sample_read = np.ascontiguousarray(sample_read, dtype="<i4")
        W, K = wl.size, x_ref.size
        assert flux.size == W and y_ref.size == K and dur_ms.size == K and sample_read.size == K
        self.ops.append(struct.pack("<i6id", 2, S, sub_scale, rng_mode, W, K, R, float(scale_factor)) + _f64(wl) + _f64(flux) +
                        _f64(x_ref) + _f64(y_ref) + _f64(dur_ms) + sample_read.tobytes())
        self.kinds.append(("plan", W))
        return len(self.ops) - 1

    def sky(self, sky_ct_s, read_dt, sky_sorted, has_sky=True):
        read_dt = np.asarray(read_dt, dtype=float)
        sky_sorted = np.ascontiguousarray(sky_sorted, dtype="<f4")
        self.ops.append(struct.pack("<idi", 3, float(sky_ct_s), read_dt.size) + _f64(read_dt) +
                        struct.pack("<ii", 1 if has_sky else 0, sky_sorted.size) + sky_sorted.tobytes())
        self.kinds.append(("sky",))
        return len(self.ops) - 1

    def alias(self, lam):
        self.ops.append(struct.pack("<id", 4, float(lam)))
        self.kinds.append(("alias",))
        return len(self.ops) - 1

    def psf(self, counts, x, y, ratio, sigl, N, rng_mode, threads_compat=1, margin=30):
        counts = np.ascontiguousarray(counts, dtype="<i4")
        n = counts.size
        self.ops.append(struct.pack("<i5i", 5, n, N, rng_mode, threads_compat, margin) + counts.tobytes() + _f64(x) + _f64(y) +
                        _f64(ratio) + _f64(sigl))
        self.kinds.append(("psf", n))
        return len(self.ops) - 1

    def run(self, variant="", timeout=600):
        exe = binary(variant)
        if not os.path.exists(exe):
            build()
        with tempfile.TemporaryDirectory() as d:
            fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
            with open(fin, "wb") as f:
                for op in self.ops:
                    f.write(op)
            env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
            env.pop("LD_PRELOAD", None)
            p = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=timeout, env=env)
            if p.returncode != 0:
                raise HarnessError("plan_harness%s exited %d:\n%s" % ("/" + variant if variant else "", p.returncode,
                                                                     p.stderr[-4000:]))
            data = open(fout, "rb").read()
        return self._decode(data)

    def _decode(self, data):
        out, pos = [], 0

        def take(fmt):
            nonlocal pos
            v = struct.unpack_from("<" + fmt, data, pos)
            pos += struct.calcsize("<" + fmt)
            return v

        def arr(dtype, n):
            nonlocal pos
            a = np.frombuffer(data, dtype=dtype, count=n, offset=pos).copy()
            pos += a.nbytes
            return a

        for kind in self.kinds:
            (code,) = take("i")
            if kind[0] == "grism":
                assert code == 1
                (ok,) = take("i")
                out.append({"table_ok": bool(ok)})
            elif kind[0] == "plan":
                assert code == 2
                r = {}
                (use,) = take("i")
                r["use_box"] = bool(use)
                r["box"] = arr("<i4", 64).reshape(16, 4)
                r["est_thrown"], r["max_chunk_electrons"], r["max_narrow"] = take("3d")
                r["n_chunks"], r["n_lane_chunks"] = take("2i")
                r["chunk_order"], r["lane_order"] = arr("u1", K_MAX_CHUNKS), arr("u1", K_MAX_CHUNKS)
                r["kb"], thin = take("2i")
                r["thin"] = bool(thin)
                r["smax"], r["wl_lo"], r["wl_hi"] = take("3d")
                (ok,) = take("i")
                r["sig_ok"] = bool(ok)
                (r["rebuilds"],) = take("q")
                W = kind[1]
                r["rate"], r["ratio"], r["sigl"] = arr("<f8", W), arr("<f8", W), arr("<f8", W)
                out.append(r)
            elif kind[0] == "sky":
                assert code == 3
                r = {}
                on, pieces = take("2i")
                r["alias_on"], r["pieces"] = bool(on), bool(pieces)
                (r["mask"],) = take("I")
                (r["L"],) = take("i")
                r["level"], r["tab0"] = arr("<f4", 16), arr("u1", 16)
                (n,) = take("i")
                r["keys"] = arr("<u4", n)
                (r["n_bg"],) = take("i")
                r["tables"] = arr("<u4", n * K_SKY_ALIAS).reshape(n, K_SKY_ALIAS) if r["alias_on"] else None
                out.append(r)
            elif kind[0] == "psf":
                assert code == 5
                (rc,) = take("i")
                (total,) = take("q")
                r = {"rc": rc, "total": total}
                if rc == 0:
                    n = kind[1]
                    r["prefix"] = arr("<u4", n + 1)
                    r["nwide"], r["nsplit"], r["nlane"] = arr("<i4", n), arr("<i4", n), arr("<i4", n)
                    a_s, a_l = take("2i")
                    r["any_split"], r["any_lane"] = bool(a_s), bool(a_l)
                    (r["run"],) = take("I")
                    r["rect"] = arr("<i4", 4)
                out.append(r)
            else:
                assert code == 4
                (fits,) = take("i")
                out.append({"fits": bool(fits), "table": arr("<u4", K_SKY_ALIAS)})
        assert pos == len(data), "undecoded output: %d of %d bytes" % (pos, len(data))
        return out


def alias_pmf(table):
    """The distribution a Walker table encodes: column = word >> 24 (uniform over 256), kept when the word's low 24
    bits are below the threshold, else the alias."""
    table = np.asarray(table, dtype=np.uint64)
    thr = (table & 0xFFFFFF).astype(np.float64) / 16777216.0
    alias = (table >> 24).astype(np.int64)
    pmf = np.zeros(K_SKY_ALIAS)
    np.add.at(pmf, np.arange(K_SKY_ALIAS), thr / K_SKY_ALIAS)
    np.add.at(pmf, alias, (1.0 - thr) / K_SKY_ALIAS)
    return pmf
