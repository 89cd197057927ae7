"""CPU: the oracle's C restatement of PSF() against the golden vectors that
the reference's own compiled C produced (scripts/make_golden_psf.py), and --
where oracle/_ref is present -- against that library directly."""
import numpy as np
import pytest

from conftest import golden_psf_cases, load_golden_psf
from oracle import clib

CASES = golden_psf_cases()


def test_golden_fixtures_present():
    assert len(CASES) >= 10


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    k = load_golden_psf(name)
    got = clib.psf_oracle(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], k["nr"], k["nc"],
                          k["test"], k["threads"])
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, k["frame"])  # bit-exact: integer counts


@pytest.mark.skipif(not clib.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("threads", [1, 2, 5, 8])
def test_oracle_matches_compiled_reference_random(threads):
    rng = np.random.RandomState(100 + threads)
    for trial in range(4):
        W = int(rng.randint(1, 400))
        N = int(rng.choice([32, 64, 100]))
        counts = rng.poisson(rng.uniform(0.2, 60), W).astype(np.int32)
        x = rng.uniform(-5, N + 5, W)
        y = rng.uniform(-5, N + 5, W)
        ratio = rng.uniform(0, 1, W)
        sl = rng.uniform(0.3, 1.0, W)
        sh = rng.uniform(3, 8, W)
        test = int(rng.randint(0, 100000))
        a = clib.psf_oracle(counts, x, y, ratio, sl, sh, N, N, test, threads)
        b = clib.psf_reference(counts, x, y, ratio, sl, sh, N, N, test, threads)
        np.testing.assert_array_equal(a, b)


def test_rand_r_restatement_matches_glibc():
    import ctypes as C
    libc = C.CDLL(None)
    libc.rand_r.argtypes = [C.POINTER(C.c_uint)]
    libc.rand_r.restype = C.c_int
    for seed in (0, 1, 25234, 25234 + 17 * 3 + 99999, 0xFFFFFFFF):
        a, b = C.c_uint(seed), C.c_uint32(seed)
        for _ in range(2000):
            assert libc.rand_r(C.byref(a)) == clib.lib().wayne_oracle_rand_r(C.byref(b))
            assert a.value == b.value


def test_never_populates_row_or_column_zero():
    # positions in (-1, 1) truncate to 0 and are rejected (pyparallel_menu.c:91-93)
    W = 50
    counts = np.full(W, 200, dtype=np.int32)
    x = np.linspace(-0.9, 0.9, W)
    y = np.full(W, 10.0)
    f = clib.psf_oracle(counts, x, y, np.zeros(W), np.full(W, 0.05), np.full(W, 1.0), 32, 32, 1, 1).reshape(32, 32)
    assert f[:, 0].sum() == 0 and f[0, :].sum() == 0


def test_invalid_inputs_raise():
    one = np.ones(3)
    with pytest.raises(ValueError):
        clib.psf_oracle(np.array([1, -1, 2]), one, one, one, one, one, 8, 8, 0, 1)
    with pytest.raises(ValueError):  # ssum * threads overflows the reference's int
        clib.psf_oracle(np.array([2 ** 30, 2 ** 30 - 1, 0]), one, one, one, one, one, 8, 8, 0, 2)
