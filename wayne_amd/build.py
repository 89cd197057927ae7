"""Build libwayne_hip.so (gfx950 only) in-tree with hipcc.

    python -m wayne_amd.build          # build if stale
    python -m wayne_amd.build --force

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwayne_hip.so")
SOURCES = [os.path.join(CSRC, "wayne_hip.hip")]
DEPS = SOURCES + sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")) + [
    os.path.join(ROOT, "include", "wayne_hip.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # no fused multiply-adds the source does not spell out: the CPU oracle
    # must see the same roundings (samplers.h)
    "-ffp-contract=off",
    "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function",
]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    # WAYNE_CXXFLAGS: extra compiler flags for experiments (e.g. -DWAYNE_RAMP_PF=2); not used by any shipped build
    cmd = [HIPCC] + FLAGS + os.environ.get("WAYNE_CXXFLAGS", "").split() + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


# Negative-control library of tests/test_extremes_gpu.py: the sky draw's sequential search WITHOUT its stop (the defect
# of rounds 1-3, k_ramp.h sky_draw_count_int).  Test infrastructure: built next to the CPU harnesses, loaded only through
# WAYNE_HIP_LIB by a test's child process, never by the product.
NEGCTL_SKY_LIB = os.path.join(ROOT, "tests", "native", "_build", "libwayne_hip_negctl_sky.so")


def build_variant(extra_flags, out, force=False, verbose=False):
    """The library with extra compiler flags, written to `out` (rebuilt when a source is newer)."""
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", out] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_negctl_sky(force=False, verbose=False):
    return build_variant(["-DWAYNE_NEGCTL_SKY_RUNAWAY"], NEGCTL_SKY_LIB, force, verbose)


# Negative-control library of tests/test_independence_gpu.py: stream keys that ADD element, sub-sample and exposure index
# into one counter word (philox.h) -- marginal laws intact, streams shared between neighbours of consecutive exposures.
NEGCTL_KEY_LIB = os.path.join(ROOT, "tests", "native", "_build", "libwayne_hip_negctl_key.so")


def build_negctl_key(force=False, verbose=False):
    return build_variant(["-DWAYNE_NEGCTL_ADDITIVE_KEY"], NEGCTL_KEY_LIB, force, verbose)


# Negative-control library of tests/test_visit_science_gpu.py: the production throwers drop the fraction of a pixel of
# every bin's position (common.h, bin_local) -- the gross form of the position-rounding defect that the visit-level
# measurement is there to exclude.
NEGCTL_FRACTION_LIB = os.path.join(ROOT, "tests", "native", "_build", "libwayne_hip_negctl_fraction.so")


def build_negctl_fraction(force=False, verbose=False):
    return build_variant(["-DWAYNE_NEGCTL_DROP_FRACTION"], NEGCTL_FRACTION_LIB, force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
