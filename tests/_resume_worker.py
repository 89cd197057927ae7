"""Rank process of tests/test_resume.py (CPU): the CLI's visit loop -- host half of every exposure, the three-deep
pipeline, the FITS writer pool, --resume -- over a STAND-IN for the GPU context whose "reads" are a pure function of
(visit seed, exposure index), as the real reads are.  Nothing here touches a GPU; the stand-in exists so that the
restart logic can be tested, with ranks that die, on the CPU box.

    python tests/_resume_worker.py <params.yml> <max exposures> <resume 0|1>      (RANK / WORLD_SIZE from the environment)

WAYNE_TEST_DIE_AFTER=n: this rank exits (os._exit(9), no clean-up) when its n-th file has been written, leaving a
half-written temporary file of the next one behind.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayne_amd import engine, exposure, fitsio, run_visit  # noqa: E402


class FakeCtx(object):
    def __init__(self, R, S):
        self.R, self.S, self.N = R, S, S - 10
        self.slots = {}

    def upload(self, slot, desc):
        self.slots[slot] = (int(desc.seed), int(desc.exposure_index), float(desc.sky_ct_s))

    def run(self, slot):
        pass

    def fetch_async(self, slot):
        pass

    def wait(self, slot):
        seed, index, sky = self.slots[slot]
        rng = np.random.RandomState((seed * 1000003 + index) & 0x7FFFFFFF)
        return (rng.uniform(0, 1000, (self.R + 1, self.S, self.S)) + sky).astype(np.float32)


class FakeEngine(object):
    def __init__(self, detector, NSAMP, SAMPSEQ, SUBARRAY):
        self.read_times = detector.get_read_times(NSAMP, SUBARRAY, SAMPSEQ)
        self.R = len(self.read_times)
        self.S = 1024 if SUBARRAY == 1024 else SUBARRAY + 10
        self.N = self.S - 10
        self.ctx = FakeCtx(self.R, self.S)
        self.has_dark = True
        self.SUBARRAY, self.flat_shift = SUBARRAY, 0

    def check_descriptor(self, sub_scale):
        pass


_engines = {}


def fake_get_engine(device, grism, detector, calibration, NSAMP, SAMPSEQ, SUBARRAY, *a, **k):
    key = (NSAMP, SAMPSEQ, SUBARRAY)
    if key not in _engines:
        _engines[key] = FakeEngine(detector, NSAMP, SAMPSEQ, SUBARRAY)
    return _engines[key]


def main():
    yml, max_exp, resume = sys.argv[1], int(sys.argv[2]), bool(int(sys.argv[3]))
    engine.get_engine = fake_get_engine
    die_after = int(os.environ.get("WAYNE_TEST_DIE_AFTER", "0"))
    if die_after:
        written = [0]
        real = fitsio.write_pieces

        def dying_write(path, pieces):
            if written[0] >= die_after and path.endswith("_raw.fits"):
                with open(path + fitsio.PART_SUFFIX, "wb") as f:     # the file it was in the middle of
                    f.write(b"SIMPLE  =                    T" + b" " * 50)
                os._exit(9)
            real(path, pieces)
            if path.endswith("_raw.fits"):
                written[0] += 1
        fitsio.write_pieces = dying_write
        os.environ["WAYNE_FITS_THREADS"] = "1"           # (one writer: "the n-th file" is well defined)
    # the direct image is a GPU-free gaussian already; sim_time is wall clock: pinned so that files can be compared
    orig_header = exposure.Exposure.generate_science_header

    def header(self, ldcoeffs=None):
        self.exp_info["sim_time"] = 0.0
        return orig_header(self, ldcoeffs=ldcoeffs)
    exposure.Exposure.generate_science_header = header
    argv = ["-p", yml, "--max-exposures", str(max_exp)] + (["--resume"] if resume else [])
    obs = run_visit.run(argv)
    print("skipped %s" % ",".join(str(i) for i in obs.skipped), flush=True)


if __name__ == "__main__":
    main()
