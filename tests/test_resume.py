"""CPU: a visit restarted after a rank died (`run_visit --resume`) generates only the files that are missing, and the
visit's files are then those of an uninterrupted run.

The visit loop is the product's (wayne_amd/run_visit.py, Observation.run_observation: host half of every exposure on a
producer thread, three exposures in flight, FITS writer pool, files written under a temporary name and renamed); the
GPU context is replaced by a stand-in whose reads are a pure function of (visit seed, exposure index) -- as the real
reads are, which is what makes a restart exact (tests/_resume_worker.py).  The same flag on the real device:
tests/test_visit_driver.py::test_cli_resume_on_the_device.
Reference: observation.py:403-405 (the exposure loop), :427 (file names), exposure.py:211-213 (delete and rewrite: the
reference has no restart; its exposures share one global numpy stream)."""
import os
import shutil
import subprocess
import sys

import pytest

from wayne_amd import fitsio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINI = os.path.join(ROOT, "tests", "fixtures", "mini_visit")
WORKER = os.path.join(ROOT, "tests", "_resume_worker.py")
N_EXP, WORLD = 12, 4


def make_visit(dst):
    """The mini visit (tests/fixtures/mini_visit: 128 x 128, NSAMP 4) stretched to N_EXP exposures."""
    import numpy as np
    shutil.copytree(MINI, dst)
    rows = {"jd": 2456196.22836 + 0.0015 * np.arange(N_EXP), "sky": 5.0 + 0.1 * np.arange(N_EXP),
            "xref": 460.0 + 0.01 * np.arange(N_EXP), "yref": 482.0 + 0.02 * np.arange(N_EXP)}
    for name, v in rows.items():
        np.savetxt(os.path.join(dst, name + ".txt"), v, fmt="%.10f")
    return os.path.join(dst, "params.yml")


def start(yml, rank, resume, die_after=0):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(WORLD), PYTHONPATH=ROOT)
    env.pop("WAYNE_TEST_DIE_AFTER", None)
    if die_after:
        env["WAYNE_TEST_DIE_AFTER"] = str(die_after)
    return subprocess.Popen([sys.executable, WORKER, yml, str(N_EXP), "1" if resume else "0"], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def run_all(yml, resume, die=None):
    procs = [start(yml, r, resume, (die or {}).get(r, 0)) for r in range(WORLD)]
    out = [p.communicate(timeout=600) for p in procs]
    return [p.returncode for p in procs], [o[0] for o in out], [o[1] for o in out]


def masked(path):
    """The file's bytes with the value of the DATE card (the day the file was written) blanked."""
    b = bytearray(open(path, "rb").read())
    i = b.find(b"DATE    = ")
    assert 0 <= i < 2880 * 4
    b[i + 10:i + 80] = b" " * 70
    return bytes(b)


def skipped_of(stdout):
    line = [l for l in stdout.splitlines() if l.startswith("skipped ")][-1]
    return [int(x) for x in line[len("skipped "):].split(",") if x]


def test_resume_after_a_dead_rank_completes_only_the_missing_files(tmp_path):
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    make_visit(a)
    make_visit(b)
    # the uninterrupted visit
    codes, outs, errs = run_all(os.path.join(a, "params.yml"), resume=False)
    assert codes == [0] * WORLD, errs
    names = ["0000_flt.fits"] + ["%04d_raw.fits" % n for n in range(1, N_EXP + 1)]
    assert sorted(os.listdir(os.path.join(a, "out"))) == sorted(names + ["params.yml", "visit_plan.txt"])
    # the same visit with rank 1 dying after its first file (of 3: exposures 1, 5, 9 -> files 0002, 0006, 0010)
    codes, outs, errs = run_all(os.path.join(b, "params.yml"), resume=False, die={1: 1})
    assert codes[1] == 9 and [codes[r] for r in (0, 2, 3)] == [0, 0, 0], (codes, errs)
    left = sorted(os.listdir(os.path.join(b, "out")))
    assert "0002_raw.fits" in left and "0006_raw.fits" not in left and "0010_raw.fits" not in left
    assert "0006_raw.fits.part" in left                                    # what it was writing when it died
    # ... and a file under its final name that something truncated (not this writer: it renames finished files)
    victim = os.path.join(b, "out", "0003_raw.fits")
    whole = os.path.getsize(victim)
    with open(victim, "r+b") as f:
        f.truncate(whole // 2)
    assert fitsio.scan(victim) is None and len(fitsio.scan(os.path.join(b, "out", "0002_raw.fits"))) == 1 + 5 * 4
    # ... and one that lost whole trailing HDUs: still a FITS file, no longer this exposure's
    victim2 = os.path.join(b, "out", "0012_raw.fits")
    with open(victim2, "r+b") as f:
        f.truncate(os.path.getsize(victim2) - 2880 * 3)
    assert len(fitsio.scan(victim2)) == 1 + 5 * 4 - 3
    before = {n: os.stat(os.path.join(b, "out", n)).st_mtime_ns for n in left if n.endswith("_raw.fits")}
    # the restart: every rank again, with --resume
    codes, outs, errs = run_all(os.path.join(b, "params.yml"), resume=True)
    assert codes == [0] * WORLD, errs
    skipped = {r: skipped_of(outs[r]) for r in range(WORLD)}
    # (0-based indices; rank 2's exposure 2 and rank 3's exposure 11 are the two damaged files)
    assert skipped == {0: [0, 4, 8], 1: [1], 2: [6, 10], 3: [3, 7]}
    after = sorted(os.listdir(os.path.join(b, "out")))
    assert after == sorted(names + ["params.yml", "visit_plan.txt"])        # complete, no temporary file left
    for n, t in before.items():
        if n not in ("0003_raw.fits", "0012_raw.fits"):
            assert os.stat(os.path.join(b, "out", n)).st_mtime_ns == t, "%s was rewritten" % n
    assert os.stat(victim).st_mtime_ns != before["0003_raw.fits"] and os.path.getsize(victim) == whole
    # the restarted visit's files are the uninterrupted visit's
    for n in names:
        assert masked(os.path.join(a, "out", n)) == masked(os.path.join(b, "out", n)), n
    # a second restart finds nothing to do
    codes, outs, errs = run_all(os.path.join(b, "params.yml"), resume=True)
    assert codes == [0] * WORLD and sum(len(skipped_of(o)) for o in outs) == N_EXP


def test_a_file_of_another_visit_is_not_taken_for_this_one(tmp_path):
    # same file name, another start time (another visit in the same directory): regenerated, not skipped
    import yaml
    a = str(tmp_path / "a")
    yml = make_visit(a)
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "WAYNE_TEST_DIE_AFTER"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, WORKER, yml, "3", "0"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cfg = yaml.safe_load(open(yml))
    jd = os.path.join(a, cfg["observation"]["exp_start_times"])
    lines = open(jd).read().split()
    lines[1] = repr(float(lines[1]) + 0.0005)                   # exposure 2 of the "new" visit starts 43 s later
    open(jd, "w").write("\n".join(lines) + "\n")
    r = subprocess.run([sys.executable, WORKER, yml, "3", "1"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert skipped_of(r.stdout) == [0, 2]


def test_scan_walks_a_file_without_reading_its_data(tmp_path):
    import numpy as np
    p = str(tmp_path / "x.fits")
    fitsio.write(p, [fitsio.HDU(fitsio.Header([("NSAMP", 3, "")]), None),
                     fitsio.HDU(fitsio.Header([]), np.arange(12, dtype=np.float64).reshape(3, 4), name="SCI"),
                     fitsio.HDU(fitsio.Header([]), None, name="ERR")])
    s = fitsio.scan(p)
    assert [size for _, size in s] == [0, 96, 0] and s[0][0]["NSAMP"] == 3 and not os.path.exists(p + ".part")
    raw = open(p, "rb").read()
    # cut inside a header, inside a payload, or with bytes after the last HDU: not a whole file; cut at an HDU
    # boundary: a whole file of fewer HDUs (the caller counts them)
    for cut in (1, 2880 + 100, 2 * 2880 + 50, len(raw) + 10):
        q = str(tmp_path / ("cut%d.fits" % cut))
        open(q, "wb").write(raw[:cut] if cut <= len(raw) else raw + b"\x00" * 10)
        assert fitsio.scan(q) is None, cut
    q = str(tmp_path / "two.fits")
    open(q, "wb").write(raw[:len(raw) - 2880])
    assert len(fitsio.scan(q)) == 2
    assert fitsio.scan(str(tmp_path / "missing.fits")) is None


def test_a_failed_write_leaves_no_file_under_the_final_name(tmp_path, monkeypatch):
    # files are written under `<name>.part` and renamed when finished: a write that dies half way (a full disk, a killed
    # process) never leaves a partial file under the name a reader -- or `--resume` -- looks for, and an older file of
    # that name stays what it was
    import numpy as np
    p = str(tmp_path / "x.fits")
    hdus = [fitsio.HDU(fitsio.Header([("NSAMP", 3, "")]), None),
            fitsio.HDU(fitsio.Header([]), np.arange(1000, dtype=np.float64).reshape(25, 40), name="SCI")]
    fitsio.write(p, hdus)
    good = open(p, "rb").read()
    real = fitsio._write_all

    def dies(fd, pieces):
        os.write(fd, bytes(pieces[0])[:1000])                 # a little of the file, then the failure
        raise OSError(28, "No space left on device")
    monkeypatch.setattr(fitsio, "_write_all", dies)
    with pytest.raises(OSError):
        fitsio.write(p, hdus)
    assert open(p, "rb").read() == good                        # the file of that name is untouched
    q = str(tmp_path / "y.fits")
    with pytest.raises(OSError):
        fitsio.write(q, hdus)
    assert not os.path.exists(q) and os.path.exists(q + fitsio.PART_SUFFIX)
    monkeypatch.setattr(fitsio, "_write_all", real)
    fitsio.write(q, hdus)
    assert open(q, "rb").read() == good and not os.path.exists(q + fitsio.PART_SUFFIX)
