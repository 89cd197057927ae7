"""GPU: the HIP electron thrower through the C ABI (wayne_psf_apply) and the
apply_psf drop-in, against the reference's golden frames and the oracle."""
import numpy as np
import pytest

from conftest import golden_psf_cases, load_golden_psf
from oracle import clib
from wayne_amd import _lib

pytestmark = pytest.mark.gpu
CASES = golden_psf_cases()


@pytest.mark.parametrize("name", CASES)
def test_replay_mode_is_bit_exact_against_reference_golden(gpu_ctx, name):
    k = load_golden_psf(name)
    got = gpu_ctx.psf_apply(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], k["nr"], k["nc"],
                            k["test"], k["threads"], rng_mode=0)
    # Integer electron counts: bit-exact.  A flip needs A*sigma+x within ~1 ulp
    # of an integer where device libm and glibc round differently (p ~ 1e-13
    # per electron); any such electron is counted here, not hidden.
    diff = int(np.abs(got.astype(np.int64) - k["frame"].astype(np.int64)).sum())
    assert got.sum() == k["frame"].sum()
    assert diff == 0, "%d electrons landed in a different pixel" % (diff // 2)


@pytest.mark.parametrize("threads", [1, 3, 7])
def test_replay_mode_random_inputs_against_oracle(gpu_ctx, threads):
    rng = np.random.RandomState(7 + threads)
    for trial in range(6):
        W = int(rng.randint(1, 3000))
        N = int(rng.choice([32, 64, 256, 512]))
        counts = rng.poisson(rng.uniform(0.1, 80), W).astype(np.int32)
        x = np.sort(rng.uniform(-8, N + 8, W))
        y = rng.uniform(0.3, 0.7) * N + 0.01 * x
        ratio = rng.uniform(0, 1, W)
        sl, sh = rng.uniform(0.3, 1.0, W), rng.uniform(3, 8, W)
        test = int(rng.randint(0, 100000))
        want = clib.psf_oracle(counts, x, y, ratio, sl, sh, N, N, test, threads)
        got = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, test, threads, rng_mode=0)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("case", ["on_the_grid", "tiny_sigma", "huge_sigma", "many_electrons", "with_reference_c"])
def test_replay_mode_where_the_float32_path_must_hand_over(gpu_ctx, case):
    # the replay thrower evaluates a position in float32 and keeps the result only outside a band around the pixel
    # boundaries (k_throw.h); these inputs sit on the band's edges: bin positions exactly on pixel corners and centres
    # (a symmetric PSF then puts half of the electrons within rounding of a boundary... of a coordinate), sigmas from
    # 1e-4 px (every electron inside the band of its bin's pixel: all fp64) to 300 px (most electrons off the frame,
    # float32 spacing of the offsets ~1e-5 px), and 2e7 electrons in a few bins (long rand_r streams, the partition of
    # 7 emulated threads).  Frames must equal the oracle's -- and the compiled reference C's -- bit for bit
    rng = np.random.RandomState(11)
    N, threads = 128, 7
    if case == "on_the_grid":
        W = 400
        x = rng.randint(2, N - 2, W).astype(float) + rng.choice([0.0, 0.5, 1.0 - 2 ** -40, 2 ** -40], W)
        y = rng.randint(2, N - 2, W).astype(float) + rng.choice([0.0, 0.5], W)
        sl, sh = rng.uniform(0.3, 1.0, W), rng.uniform(3, 8, W)
        counts = rng.poisson(300, W)
    elif case == "tiny_sigma":
        W = 300
        x, y = rng.uniform(1, N - 1, W), rng.uniform(1, N - 1, W)
        x[::3] = np.round(x[::3])                       # on a boundary with a PSF of 1e-4 px: both sides get electrons
        sl, sh = np.full(W, 1e-4), rng.choice([1e-4, 1e-3, 0.02], W)
        counts = rng.poisson(500, W)
    elif case == "huge_sigma":
        W = 200
        x, y = rng.uniform(-50, N + 50, W), rng.uniform(-50, N + 50, W)
        sl, sh = rng.uniform(20, 60, W), rng.uniform(100, 300, W)
        counts = rng.poisson(2000, W)
    elif case == "many_electrons":
        W = 5
        x, y = rng.uniform(40, 90, W), rng.uniform(40, 90, W)
        sl, sh = rng.uniform(0.4, 0.9, W), rng.uniform(4, 7, W)
        counts = np.array([4_000_000, 1, 9_000_000, 0, 7_000_000])
    else:
        if not clib.have_ref():
            pytest.skip("oracle/_ref not built")
        W = 1500
        x = np.sort(rng.uniform(5, N - 5, W))
        y = 0.5 * N + 0.01 * x
        sl, sh = rng.uniform(0.4, 0.9, W), rng.uniform(4, 7, W)
        counts = rng.poisson(900, W)
    ratio = rng.uniform(0, 1, W)
    counts = counts.astype(np.int32)
    got = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, 4242, threads, rng_mode=0)
    if case == "with_reference_c":
        want = clib.psf_reference(counts, x, y, ratio, sl, sh, N, N, 4242, threads)
    else:
        want = clib.psf_oracle(counts, x, y, ratio, sl, sh, N, N, 4242, threads)
    np.testing.assert_array_equal(got, want)
    assert got.sum() > 0


def test_replay_soak_against_the_restatement_and_the_reference_c(gpu_ctx):
    # a fixed-seed stretch of scripts/soak_replay.py: random bins, frames from 16 to 1014 pixels, PSF widths from 1e-3 to
    # 200 px, 1-16 emulated threads, positions on and off the frame and on pixel boundaries, a bin of 10^5-10^6 electrons
    # now and then -- every frame equal to the CPU restatement's, every fifth also to the compiled reference C's
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import soak_replay
    bad, electrons = soak_replay.run(gpu_ctx, 60, 4)
    assert bad == 0 and electrons > 1e6


def test_apply_psf_dropin_signature_and_dtype(gpu_ctx):
    from wayne_amd import pyparallel
    k = load_golden_psf("s64_t3")
    out = pyparallel.apply_psf(k["counts"].astype(np.float64), k["x"], k["y"], k["ratio"], k["sl"], k["sh"],
                               k["nr"], k["nc"], k["test"], k["threads"])
    assert out.dtype == np.float64 and out.shape == (k["nr"] * k["nc"],)   # pyparallel.pyx:31-34
    np.testing.assert_array_equal(out, k["frame"].astype(np.float64))


def test_philox_mode_against_oracle_same_counters(gpu_ctx):
    # Same Philox counters, fp32 Box-Muller: the device uses the hardware
    # sin/cos/log2 units, the oracle libm, so a small fraction of electrons may
    # truncate into the neighbouring pixel.  Totals must agree exactly.
    k = load_golden_psf("s256_t4")
    for seed, exp, sub in [(1963, 0, 0), (7, 12, 3)]:
        want = clib.psf_philox_oracle(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], 256, 256,
                                      seed, exp, sub)
        got = gpu_ctx.psf_apply(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], 256, 256, seed,
                                threads=1, rng_mode=1, exposure=exp, subsample=sub)
        moved = int(np.abs(got.astype(np.int64) - want.astype(np.int64)).sum()) // 2
        total = int(want.sum())
        assert abs(int(got.sum()) - total) <= 2          # only edge-of-frame flips change the total
        assert moved <= 2e-3 * total, "%d of %d electrons moved" % (moved, total)


def test_philox_mode_is_deterministic_and_geometry_invariant(gpu_ctx):
    import os
    k = load_golden_psf("s1014_t4")
    a = gpu_ctx.psf_apply(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], 1014, 1014, 42, rng_mode=1)
    b = gpu_ctx.psf_apply(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], 1014, 1014, 42, rng_mode=1)
    np.testing.assert_array_equal(a, b)
    gpu_ctx.set_knob("tile_ints", 300)             # tiny LDS tile: most electrons take the global path
    try:
        c = gpu_ctx.psf_apply(k["counts"], k["x"], k["y"], k["ratio"], k["sl"], k["sh"], 1014, 1014, 42, rng_mode=1)
    finally:
        gpu_ctx.set_knob("tile_ints", None)
    np.testing.assert_array_equal(a, c)
    assert a.sum() == k["counts"].sum()             # nothing falls off this frame


def test_philox_mode_psf_moments(gpu_ctx):
    # one bin, many electrons: the double gaussian's second moment
    n, N = 400000, 128
    ratio, sl, sh = 0.25, 0.7, 5.5
    f = gpu_ctx.psf_apply([n], [64.5], [64.5], [ratio], [sl], [sh], N, N, 3, rng_mode=1).reshape(N, N)
    assert f.sum() == n
    ys, xs = np.mgrid[0:N, 0:N]
    mx, my = (f * xs).sum() / n, (f * ys).sum() / n
    # truncation toward zero of x ~ N(64.5, s): pixel = floor(x), mean 64.0
    assert abs(mx - 64.0) < 0.05 and abs(my - 64.0) < 0.05
    var = (f * (xs - mx) ** 2).sum() / n
    want = ratio * sh ** 2 + (1 - ratio) * sl ** 2 + 1.0 / 12.0
    assert abs(var - want) < 0.03 * want


def test_errors(gpu_ctx):
    from wayne_amd import _lib
    one = np.ones(3)
    with pytest.raises(_lib.WayneError) as e:
        gpu_ctx.psf_apply([1, -1, 2], one, one, one, one, one, 8, 8, 0, 1)
    assert e.value.status == _lib.E_NEGATIVE
    with pytest.raises(_lib.WayneError) as e:
        gpu_ctx.psf_apply([2 ** 30, 2 ** 30 - 1, 0], one, one, one, one, one, 8, 8, 0, 2)
    assert e.value.status == _lib.E_OVERFLOW
    with pytest.raises(_lib.WayneError):
        gpu_ctx.psf_apply([1, 1, 1], one, one, one, one, one, 8, 16, 0, 1)
    # empty input -> all-zero frame
    z = gpu_ctx.psf_apply(np.zeros(0, np.int32), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0),
                          16, 16, 0, 1)
    assert z.shape == (256,) and not z.any()


def test_hostile_inputs_all_modes(gpu_ctx):
    # NaN / infinite / far-away positions, zero and negative sigmas, ratios outside [0, 1]: the reference's C
    # turns all of these into rejected electrons (the (int) of a non-finite double fails 0 < pos < n,
    # pyparallel_menu.c:91-93).  Replay mode must agree with the oracle bit for bit; the Philox modes
    # must stay on the frame and conserve what the oracle keeps, whatever they are fed.
    rng = np.random.default_rng(17)
    W, N = 600, 128
    counts = rng.integers(0, 400, W).astype(np.int32)
    counts[::5] = rng.integers(0, 12, counts[::5].size)
    x = rng.uniform(-30, N + 30, W)
    y = rng.uniform(-30, N + 30, W)
    ratio = rng.uniform(-0.5, 1.5, W)
    sl = rng.uniform(0.0, 1.2, W)
    sh = rng.uniform(0.0, 7.0, W)
    x[3], y[4] = np.nan, np.nan
    x[10], y[11] = np.inf, -np.inf
    x[20], y[21] = 1e30, -1e30
    sl[30], sh[31] = -0.7, -3.0
    sl[40] = 0.0
    ratio[50] = np.nan
    # sigmas that are not finite: the reference's (int) of a non-finite position keeps none of the electrons that take
    # them (pyparallel_menu.c:91-93) -- the wide ones of bin 60 / 62, the narrow ones of bin 61 / 63, everything of 64 --
    # while the bin's other electrons land as usual (the production throwers settle this once per bin: k_narrow.h bad_h / bad_l)
    for b in (60, 61, 62, 63, 64):
        x[b], y[b], ratio[b], counts[b] = 40.0 + b, 50.5, 0.5, 300
    sh[60], sl[61], sh[62], sl[63] = np.nan, np.nan, np.inf, -np.inf
    sl[64] = sh[64] = np.nan
    want = clib.psf_oracle(counts, x, y, ratio, sl, sh, N, N, 77, 3)
    got = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, 77, threads=3, rng_mode=_lib.RNG_REPLAY)
    np.testing.assert_array_equal(got, want)
    f = want.reshape(N, N)
    assert f[0].sum() == 0 and f[:, 0].sum() == 0
    for mode in (_lib.RNG_PHILOX, _lib.RNG_SPLIT):
        a = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, 5, rng_mode=mode)
        b = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, 5, rng_mode=mode)
        np.testing.assert_array_equal(a, b)
        fa = a.reshape(N, N)
        assert a.min() >= 0 and fa[0].sum() == 0 and fa[:, 0].sum() == 0
        assert abs(int(a.sum()) - int(want.sum())) < 6 * np.sqrt(want.sum())      # same loss off the frame
    # and the default mode against its own oracle on the same counters
    w2 = clib.psf_split_oracle(counts, x, y, ratio, sl, sh, N, 5, 0, 0)
    a = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, 5, rng_mode=_lib.RNG_SPLIT)
    moved = int(np.abs(a.astype(np.int64) - w2).sum()) // 2
    assert moved <= 5 + 1e-3 * w2.sum(), "%d of %d electrons moved" % (moved, w2.sum())
    # the five bins alone: exactly the electrons with a finite sigma arrive (150 of 300 each, none of bin 64's), in all modes
    sel = np.zeros(W, dtype=bool)
    sel[60:65] = True
    c5 = np.where(sel, counts, 0).astype(np.int32)
    for mode in (_lib.RNG_REPLAY, _lib.RNG_PHILOX, _lib.RNG_SPLIT):
        f5 = gpu_ctx.psf_apply(c5, x, y, ratio, sl, sh, N, N, 9, threads=2, rng_mode=mode)
        assert int(f5.sum()) == 4 * 150, (mode, int(f5.sum()))


def test_inner_boundary_from_plain_c(tmp_path):
    # examples/psf_from_c.c: the thrower called from a C program that sees nothing but include/wayne_hip.h and
    # libwayne_hip.so (no Python, no torch on its side of the boundary) -- a golden vector of the reference in, the
    # reference's frame out, bit for bit
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "wayne_amd")
    exe = str(tmp_path / "psf_from_c")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "psf_from_c.c"), "-o", exe, "-L", libdir, "-lwayne_hip",
                    "-Wl,-rpath," + libdir], check=True)
    for name in ("s64_t3", "s256_t4"):
        k = load_golden_psf(name)
        n = k["counts"].size
        with open(str(tmp_path / "in.bin"), "wb") as f:
            f.write(np.array([n, k["nr"], k["nc"], k["test"], k["threads"]], dtype=np.int32).tobytes())
            f.write(k["counts"].astype(np.int32).tobytes())
            for a in ("x", "y", "ratio", "sl", "sh"):
                f.write(np.ascontiguousarray(k[a], dtype=np.float64).tobytes())
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        frame = np.fromfile(str(tmp_path / "out.bin"), dtype=np.int32)
        np.testing.assert_array_equal(frame, k["frame"].ravel())
        assert ("%d electrons" % int(k["frame"].sum())) in r.stdout


def test_a_bins_fraction_of_a_pixel_survives_at_the_far_side_of_the_frame(gpu_ctx):
    # The reference adds an electron's offset to its bin's position in fp64 (pyparallel_menu.c:91-92).  Through round 5
    # the production throwers cast the FRAME coordinate to float32 first: at x >= 512 that rounds to 6.1e-5 px, so a bin
    # 2e-5 px below a pixel boundary was moved ONTO it and its electrons into the next pixel.  With a PSF far narrower
    # than that distance the law is a certainty -- every electron in pixel (floor(y), floor(x)) -- and all three modes
    # must give the same frame: the bit-exact replay of the reference, and the production modes from bin-local
    # coordinates (wayne_amd/csrc/common.h, bin_local).  No oracle in between.
    N = 1014
    ks = np.arange(520, 1010, 7)
    eps = 2e-5
    x = np.concatenate([ks + 1.0 - eps, ks + eps, ks + 0.5])                 # just below, just above a boundary, mid-pixel
    y = np.concatenate([ks[::-1] + eps, ks[::-1] + 1.0 - eps, ks[::-1] + 1.0 - eps])
    n = x.size
    counts = np.full(n, 300, dtype=np.int32)
    counts[::3] = 20                                                          # thin bins too
    ratio = np.full(n, 0.25)                                                  # a quarter of each bin takes sigma_h
    sl, sh = np.full(n, 1e-6), np.full(n, 2e-6)
    want = np.zeros((N, N), dtype=np.int64)
    np.add.at(want, (np.floor(y).astype(int), np.floor(x).astype(int)), counts)
    assert np.all(np.float32(x[:ks.size]) == ks + 1.0)                        # the cast that used to be made: ON the boundary
    for mode in (_lib.RNG_REPLAY, _lib.RNG_PHILOX, _lib.RNG_SPLIT):
        got = gpu_ctx.psf_apply(counts, x, y, ratio, sl, sh, N, N, 11, threads=3, rng_mode=mode).reshape(N, N)
        np.testing.assert_array_equal(got, want, err_msg="rng_mode %d" % mode)
