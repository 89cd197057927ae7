"""CPU: libwayne_hip.so builds, loads and exports every symbol that
include/wayne_hip.h declares; without a GPU the path fails loudly."""
import os
import re

import numpy as np
import pytest

from wayne_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "wayne_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wayne_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libwayne_hip.so does not export %s" % n
    # and the binding covers them all
    assert set(names) == set(_lib.SYMBOLS)


def test_abi_version_and_strerror():
    L = _lib.load()
    assert L.wayne_abi_version() == 7 == _lib.ABI_VERSION
    assert L.wayne_strerror(0) == b"ok"
    assert b"gfx950" in L.wayne_strerror(_lib.E_NODEVICE)
    assert L.wayne_build_flags() == b""            # the shipped library carries no negative-control / timing switch


def test_the_environment_is_read_in_one_place_only():
    # the tuning knobs are frozen when a context is created (wayne_hip.h, wayne_ctx_set_knob): no entry point of a live
    # context may look at the environment -- one getenv in the whole library, inside wayne_ctx_create
    src = open(os.path.join(ROOT, "wayne_amd", "csrc", "wayne_hip.hip")).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert len(re.findall(r"\bgetenv\s*\(", code)) == 1
    create = code[code.index("wayne_ctx* wayne_ctx_create("):code.index("void wayne_ctx_destroy(")]
    assert "getenv" in create
    for h in os.listdir(os.path.join(ROOT, "wayne_amd", "csrc")):
        if h.endswith(".h"):
            assert "getenv" not in open(os.path.join(ROOT, "wayne_amd", "csrc", h)).read(), h
    # every knob the header names is one the binding knows, and the other way round
    header = open(os.path.join(ROOT, "include", "wayne_hip.h")).read()
    named = header[header.index("Names:"):header.index("None changes a frame")]
    names = set(re.findall(r"\b([a-z]+(?:_[a-z]+)+|batch|thin|streams)\b", named)) - {"timing", "builds", "only"}
    assert names == set(_lib.KNOBS), names ^ set(_lib.KNOBS)
    table = set(re.findall(r'\{"([a-z_]+)", "WAYNE_[A-Z_]+", &Knobs::', src))
    assert table == set(_lib.KNOBS)


def test_a_flagged_library_is_refused(tmp_path, monkeypatch):
    # a negative-control or timing build names its switches in wayne_build_flags(); the binding refuses it unless the
    # caller says it means to load one
    import subprocess
    import sys
    lib = os.path.join(ROOT, "tests", "native", "_build", "libwayne_hip_negctl_sky.so")
    if not os.path.exists(lib):
        pytest.skip("negative-control library not built")
    code = "from wayne_amd import _lib; _lib.load(); print(_lib.library_info()[1])"
    env = dict(os.environ, WAYNE_HIP_LIB=lib)
    env.pop("WAYNE_ALLOW_FLAGGED_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WAYNE_NEGCTL_SKY_RUNAWAY" in r.stderr
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, WAYNE_ALLOW_FLAGGED_LIB="1"),
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "WAYNE_NEGCTL_SKY_RUNAWAY"


def test_host_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in _lib.philox4x32(ctr, key)) == want


def test_struct_layouts_match_header_sizes():
    # a drifted struct would corrupt every call: check the sizes the C side sees
    import ctypes as C
    assert C.sizeof(_lib.GrismDesc) == 8 * (9 + 9 + 12) + 8 + 16 + 16
    assert C.sizeof(_lib.Calibration) == 8 + 8 * (4 + 1 + 1 + 4 + 1 + 1 + 1)


@pytest.mark.skipif(_lib.device_count() > 0, reason="a GPU is present")
def test_no_gpu_fails_loudly():
    with pytest.raises(_lib.WayneNoDeviceError):
        _lib.Context(0)
    from wayne_amd import pyparallel
    with pytest.raises(_lib.WayneError):
        pyparallel.apply_psf(np.ones(3), np.ones(3), np.ones(3), np.ones(3), np.ones(3), np.ones(3), 8, 8, 0, 1)


def test_header_is_plain_c_and_the_binding_mirrors_its_layout(tmp_path):
    # include/wayne_hip.h is the boundary a C (cgo / JNI / ctypes) caller binds: it must compile as strict C99 and as
    # C++ by itself, and the ctypes mirror must agree with the compiler on the size of every struct and the offset of
    # every field (a drifted struct corrupts every call silently)
    import ctypes as C
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    pairs = [("wayne_grism_desc", _lib.GrismDesc), ("wayne_calibration", _lib.Calibration),
             ("wayne_exposure_desc", _lib.ExposureDesc), ("wayne_profile", _lib.Profile)]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "wayne_hip.h"', "int main(void) {"]
    for cname, ct in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in ct._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = str(tmp_path / "layout")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", exe], check=True)
    out = dict(l.split() for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in pairs:
        assert int(out[cname]) == C.sizeof(ct), cname
        for f in ct._fields_:
            assert int(out["%s.%s" % (cname, f[0])]) == getattr(ct, f[0]).offset, "%s.%s" % (cname, f[0])
    gxx = shutil.which("g++")
    if gxx:
        subprocess.run([gxx, "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-x", "c++", "-I", inc, "-c", str(src),
                        "-o", str(tmp_path / "layout.o")], check=True)
