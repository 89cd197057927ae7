import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The checker is built BEFORE the test modules are collected: several of them decide at import time whether the compiled
# reference kernel is there (`skipif(not clib.have_ref())`), and in a fresh clone it is only there once oracle/Makefile has
# run (a no-op when the libraries are up to date; oracle/_ref needs /root/reference and is skipped without it).
from oracle import clib as _clib  # noqa: E402
_clib.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_psf_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "psf_*.npz")))


def load_golden_psf(name):
    z = np.load(os.path.join(GOLDEN, "psf_%s.npz" % name))
    k = {n: z[n] for n in z.files}
    for n in ("nr", "nc", "test", "threads"):
        k[n] = int(k[n])
    frame = np.zeros(k["nr"] * k["nc"], dtype=np.int32)
    frame[k["idx"]] = k["val"]
    k["frame"] = frame
    return k


@pytest.fixture(autouse=True)
def _knobs_back_to_default():
    """A test that sets tuning knobs on the live contexts (_lib.set_knob_all) leaves none behind, pass or fail."""
    yield
    from wayne_amd import _lib
    if _lib._knob_defaults:
        _lib.reset_knobs_all()


@pytest.fixture(scope="session")
def gpu_ctx():
    """One wayne_ctx on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    from wayne_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
