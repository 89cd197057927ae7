#!/usr/bin/env python3
"""bench.py -- simulated WFC3-IR exposures/sec on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4]

One "step" = one whole exposure (1014x1014 frame, NSAMP = 16, 128-sub-sample
spatial scan, 1e9 electrons: BASELINE.json configs[3], the configuration the
metric is quoted on) synthesised by the HIP path: k_prep_wl, k_prep_sub,
k_prep_fix (+ cosmic-ray hits), k_lane, k_narrow (+ k_throw for oversized bins), k_ramp.  All inputs and calibration planes are
resident in HBM before the timed region; outputs stay in HBM (device-complete
rate).  Exposures are independent: with N ranks each rank runs its own K
exposures (round-robin exposure indices, no collective in the data path) ->
weak scaling.

Launching.  Under torchrun (RANK / WORLD_SIZE set) this process is one rank.
Without it, `--gpus N` with N > 1 makes THIS process a launcher: before torch
or HIP is touched it starts N fresh child processes (one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set), waits for them and relays rank 0's
JSON line; a line that does not report N ranks is an error.  The ranks meet
over gloo for the barrier and the max-over-ranks of the elapsed time -- there
is nothing for RCCL to do in this path.

The timed region is repeated REPS times (each repetition = exactly K steps
bracketed by a barrier + device synchronisation, max over ranks); `value` is
the median repetition, the others are listed in `repetitions`.

Prints ONE JSON line on rank 0 (the driver's contract) with extra objects:
  roofline      the fused up-the-ramp kernel k_ramp against the HBM roof
  cpu_baseline  the reference's C thrower + the numpy restatement of the host
                loop, timed on this box's host cores on a bounded sample
  delivered / end_to_end
                SURVEY 8(d)'s second leg at every N: reads in pinned host memory
                (PCIe-inclusive), barrier + synchronise either side, max over ranks,
                every rank's own rate listed -- never `value`
  per_electron / replay_bit_exact / out_f64 / two_streams / psf_apply_replay   (N = 1)
                the same workload measured like-for-like with the reference
                (every electron thrown; float64 reads, with a roofline block of its
                own) -- never `value`
"""
import argparse
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_GBS = 63.0         # MI355X_MICROARCH.md: PCIe Gen5 x16
REPS = 5
SUSTAIN_S = 6.5       # length of the sustained-rate pass (longer than the period of the driver's gpu_busy sampler)
# what the reads of the timed region are, as `config.workload` says it (tests/test_bench_contract.py holds the strings):
# float32 is the default of every entry point -- ExposureGenerator.scanning_frame, Observation, VisitRunner, the CLI
READS_LABEL = {False: "f32 reads (the default of ExposureGenerator, Observation, VisitRunner and the CLI)",
               True: "f64 reads (out_dtype=float64 / --float64-reads: the reference's dtype)"}
RAMP_EVENTS_EVERY = 4   # k_ramp's HIP events in the timed region: on every 4th exposure (they cost the stream ~10 us a pair)


def ramp_bytes(N, S, R, out_bytes, acc_segments=None):
    """Compulsory HBM bytes of ONE k_ramp launch as designed (DESIGN.md "k_ramp"):
    per interior pixel and read: dark SCI + ERR (4 + 4), read written (out_bytes); the int64 accumulator
    (8) only where the read's electrons can have landed -- `acc_segments` 64-accumulator segments in all
    (wayne_exposure_debug_boxes: the segments one wave loads; None: every interior pixel of every read, the
    design of rounds 1-2); once per pixel: pixel flat 4, sky 4, c1..c4 16, zero read written (out_bytes).
    Border pixels only write their reads.  The accumulators that left zero are also written back as
    zeros and cosmic-ray segments are loaded too: exposure dependent, left out here (the conservative
    choice for `roofline.achieved`; the PMC figure `roofline.traffic` contains them)."""
    inner = N * N
    acc = R * inner * 8 if acc_segments is None else int(acc_segments) * 64 * 8
    once_inner = 4 + 4 + 16
    return R * (inner * (4 + 4) + S * S * out_bytes) + acc + inner * once_inner + S * S * out_bytes


def survey_bytes(N, S, R, K, W, out_bytes, A_fp):
    """R (12 N^2 + 28 S^2 + B_out S^2) + (4 + B_out) S^2 + 8 K W + 16 A_fp
    -- the unfused reference structure (gain / sky / linearity / zero read re-read per read)."""
    return R * (12 * N * N + 28 * S * S + out_bytes * S * S) + (4 + out_bytes) * S * S + 8 * K * W + 16 * A_fp


def csrc_hash():
    """Hash of everything the HIP library is built from: profile-derived numbers are only quoted
    while the kernels they were measured on are the kernels in the tree."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "wayne_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/wayne_hip.h"]:
        p = os.path.normpath(os.path.join(d, f))
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(visit, budget_s=20.0):
    """Time the CPU structure of the reference on a bounded sample of exposure 0 of the
    workload and extrapolate to one exposure: per sub-sample one C thrower call + the
    numpy passes of _gen_subsample, per read _add_read_reductions, then the post-ramp
    stage (exposure_generator.py:336-444)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import clib, wayne_oracle as wo
    det, gr, eo = wo.from_calibration(visit.calibration, visit.grism.name, visit.NSAMP, visit.SAMPSEQ,
                                      visit.SUBARRAY)
    kind = "reference" if clib.have_ref() else "port"
    thrower = clib.psf_reference if clib.have_ref() else clib.psf_oracle
    kw = visit.frame_kwargs(0)
    draws = wo.LegacyDraws(visit.seed)
    wl = kw["wl"]
    i0, i1 = wo.crop_spectrum_ind(gr.wl_limits[0], gr.wl_limits[1], wl.copy())
    s_wl = wl[i0:i1]
    gr.set_current_wavelength_only_dependent_array(s_wl)
    mids, durs = kw["sample_mid_points"], kw["sample_durations"]
    s_y = eo._gen_sample_yref(kw["y_ref"], mids, kw["scan_speed"] / 1000.)
    sub_scale = 0 if visit.SUBARRAY == 1024 else 507 - visit.SUBARRAY // 2
    threads_used = [1]

    def throw(counts, x, y, ratio, sl, sh, ny, nx, seed, threads, k):
        return thrower(np.asarray(counts).astype(np.int32), x, y, ratio, sl, sh, ny, nx, int(seed), threads_used[0])

    def one_subsample(k):
        pixel_array = det.gen_pixel_array(visit.SUBARRAY, light_sensitive=True)
        flux = kw["stellar_flux"][i0:i1] * (1. - kw["planet_signal"][k][i0:i1])
        t = time.perf_counter()
        frame, counts, _, _ = eo._gen_subsample(kw["x_ref"], s_y[k], s_wl, flux, pixel_array, durs[k], 12345 + k, 1,
                                                kw["scale_factor"], True, True, draws, k, throw, sub_scale)
        pixel_array += frame
        return time.perf_counter() - t, float(np.sum(counts)), pixel_array

    # the fastest OpenMP team size on this box among 1, 2, 4, ... host_cpus (the reference's yml uses 4;
    # its scatter loop is serial, so more threads only speed up the normal generation)
    ncpu = os.cpu_count() or 1
    tried = {}
    th = 1
    while th <= ncpu:
        threads_used[0] = th
        dt, ne, _ = one_subsample(0)
        tried[th] = dt
        th *= 2
    best = min(tried, key=tried.get)
    threads_used[0] = best
    t_sub, n_sub, electrons, pixel_array = [], 0, 0.0, None
    t0 = time.perf_counter()
    while n_sub < visit.K and (time.perf_counter() - t0) < budget_s * 0.6:
        dt, ne, pixel_array = one_subsample(n_sub)
        t_sub.append(dt)
        electrons += ne
        n_sub += 1
    # one read's per-read stage and the post-ramp stage on two reads
    dt_read = float(np.diff(np.concatenate([[0.], eo.read_times]))[1])
    t = time.perf_counter()
    full = eo._add_read_reductions(pixel_array, dt_read, False, False, kw["sky_background"], True,
                                   kw["cosmic_rate"], draws, 0)
    t_read = time.perf_counter() - t
    reads = [eo._gen_zero_read(True), full.copy(), full.copy() * 2]
    saved_R = len(reads) - 1
    t = time.perf_counter()
    eo._post_exposure_reductions(reads, True, True, True, True, draws)
    t_post = (time.perf_counter() - t) / saved_R
    R = visit.NSAMP - 1
    per_exposure = float(np.mean(t_sub)) * visit.K + (t_read + t_post) * R
    return {"value": 1.0 / per_exposure, "unit": "exposures/s", "cores": int(best), "kind": kind,
            "sample": "%s exposure 0: %d of %d sub-samples (%.3g electrons, %.2f s each: C thrower threads=%d + "
                      "numpy passes), 1 of %d per-read stages (%.2f s), post-ramp stage on 2 reads (%.2f s/read); "
                      "extrapolated to one exposure = %.1f s" % (visit.name, n_sub, visit.K, electrons,
                                                                 float(np.mean(t_sub)), best, R, t_read,
                                                                 t_post, per_exposure),
            "host_cpus": ncpu, "threads_tried_s_per_subsample": {str(k): round(v, 3) for k, v in tried.items()},
            "thrower_electrons_per_s": electrons / float(np.sum(t_sub))}


def timed_pass(ctx, slot_of, steps, warmup, sync=None, events_every=0, device_sync=None):
    """One rank's timed pass: `warmup` untimed exposures, synchronise (`sync`: the caller's barrier + device
    synchronise; default the context's own), exactly `steps` exposures, synchronise (the context's streams, then
    `device_sync` -- bench.py passes torch.cuda.synchronize, the contract's bracket) -> elapsed seconds by this
    process's clock.  Every device-complete rate of bench.py AND of scripts/bench_configs.py goes through here."""
    sync = sync or ctx.synchronize
    for j in range(warmup):
        ctx.run(slot_of(j))
    sync()
    t0 = time.perf_counter()
    for j in range(warmup, warmup + steps):
        if events_every:
            ctx.profile_enable((j - warmup) % events_every == 0)
        ctx.run(slot_of(j))
    ctx.synchronize()
    if device_sync is not None:
        device_sync()
    return time.perf_counter() - t0


def median_of_passes(ctx, slot_of, steps, reps=3, warmup=4, sync=None):
    """Median exposures/s of `reps` passes of `steps` exposures; `warmup` untimed exposures before the first pass."""
    vals = [steps / timed_pass(ctx, slot_of, steps, warmup if rep == 0 else 0, sync) for rep in range(reps)]
    return float(np.median(vals)), vals


def dtype_label(ramp_variant, out_f64):
    """`dtype` of the JSON line: the arithmetic the TIMED kernels compute in, derived from the k_ramp instantiation the
    library reports for the timed slots (wayne_exposure_ramp_variant) -- not a fixed string."""
    thrower = "f32 thrower (int32 LDS tiles -> int64 fixed-point accumulators, 2^-28 e-)"
    if re.match(r"k_ramp<float, true, 1, false, (true|false)>$", ramp_variant):
        # (ramp_body's all-float32 chain needs SKY == 1 && !NOISE, k_ramp.h; sky drawn in pieces (2) or the gaussian
        # stage on run the fp64 cumulative sum)
        ramp = "exact integer sums (int64 accumulators + int32 sky counts) then an all-f32 per-read chain"
    elif "<float" in ramp_variant:
        ramp = "f64 cumulative sum and per-read chain, reads rounded to f32"
    else:
        ramp = "f64 cumulative sum and per-read chain"
    return "%s: %s; %s; %s reads" % (ramp_variant, ramp, thrower, "f64" if out_f64 else "f32")


# ---------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ---------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Start n fresh rank processes (nothing in THIS process has touched torch or HIP), wait for them, relay
    rank 0's JSON line.  Returns the exit code."""
    from wayne_amd import launch
    codes, out0 = launch.launch_ranks(n, [sys.executable, os.path.abspath(__file__)] + argv,
                                      extra_env={"WAYNE_BENCH_CHILD": "1"}, capture_rank0=True)
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        sys.stdout.write(out0 or "")
        return 1
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, got %d\n" % len(lines))
        return 1
    line = json.loads(lines[0])
    if line.get("n_gpus") != n or line.get("ranks_reported") != n:
        sys.stderr.write("bench.py: --gpus %d but the line reports n_gpus=%s, ranks_reported=%s\n" % (
            n, line.get("n_gpus"), line.get("ranks_reported")))
        return 1
    print(lines[0], flush=True)
    return 0


_json_out = None


def _claim_stdout():
    """stdout carries ONE line, the JSON line.  Libraries write there too -- under torchrun gloo announces "[Gloo] Rank 0 is
    connected to 1 peer ranks" on file descriptor 1 from C++ -- so the descriptor is kept aside for the line and
    everything else that is written to 1, by anyone, goes to stderr."""
    global _json_out
    if _json_out is None:
        sys.stdout.flush()
        _json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_line(obj):
    out = _json_out if _json_out is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def dry_run(rank, world):
    """`--dry-run`: the rendezvous of the N-rank path without any GPU work (gloo barrier, max-reduce, gather), so
    that the launcher and the ranks' meeting can be rehearsed at any N on a CPU.  Prints a line that cannot be
    mistaken for a measurement."""
    if os.environ.get("WAYNE_DRY_RUN_FAIL_RANK") == str(rank):     # rehearsal of a rank that dies before the rendezvous
        raise SystemExit(3)                                       # (tests/test_bench_contract.py: the launcher must fail fast)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from wayne_amd import launch
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=launch.rendezvous_timeout())
    dist.barrier()
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    got = [None] * world
    dist.all_gather_object(got, rank)
    if rank == 0:
        emit_line({"dry_run": True, "n_gpus": world, "ranks_reported": len(set(got)), "max_rank": t.item(), "value": None})
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out-f64", action="store_true", help="float64 reads (the reference's dtype) in the timed region")
    ap.add_argument("--thrower", default="split", choices=["split", "electron"],
                    help="split: narrow PSF component drawn as a multinomial (WAYNE_RNG_SPLIT, same distribution); "
                         "electron: every electron thrown individually (WAYNE_RNG_PHILOX)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="HIP streams in the timed region (1: kernels of different exposures never co-run, so "
                         "per-kernel event times are clean; the 2-stream rate is reported separately as two_streams)")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extra-pass", action="store_true",
                    help="skip the passes after the timed region (use under rocprofv3 so that the kernel statistics "
                         "cover the timed region only)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    os.environ["WAYNE_STREAMS"] = "2"        # the context always owns two streams; slots select them
    _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    n_gpus = world
    if args.dry_run:
        if world < 2:
            raise SystemExit("--dry-run rehearses the multi-rank rendezvous: use --gpus N with N > 1")
        return dry_run(rank, world)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    # one process per GPU; WAYNE_BENCH_SHARE_GPU=1 lets several ranks share device 0 (a rehearsal of the
    # multi-process path on a one-GPU box)
    share = os.environ.get("WAYNE_BENCH_SHARE_GPU") == "1"
    device = 0 if share else local_rank
    if device >= torch.cuda.device_count():
        raise SystemExit("rank %d: no GPU %d on this node (set WAYNE_BENCH_SHARE_GPU=1 to share device 0)" % (
            rank, device))
    torch.cuda.set_device(device)
    from wayne_amd import launch
    launch.pin_to_gpu_numa(device)           # the rank's host threads on its GPU's NUMA node (best effort)
    dist = None
    grp = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo: a barrier and a max-reduce of one double on the host -- the data path has no collective
        # (a rank that never arrives is an error after launch.RENDEZVOUS_TIMEOUT_S, not a wait until the driver's limit)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=launch.rendezvous_timeout())
        dist.barrier()
        # ... and only the rendezvous: the collectives of the measurement run in a group of their own whose timeout
        # does not turn one rank's slow pass (a cold page cache, a first-time build) into a lost line
        import datetime
        grp = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=30))

    from wayne_amd import _lib, calibration, detector, engine, grism, synthetic, visit as wvisit
    from wayne_amd.exposure_generator import ExposureGenerator

    cal = calibration.CalibrationSet.synthetic(11)
    det = detector.WFC3_IR()
    cfg = synthetic.CONFIGS[args.config]
    gr = grism.G141(cal) if cfg["grism"] == "G141" else grism.G102(cal)
    total = args.warmup + args.steps
    # exposure j of this rank is exposure index rank + j * n_gpus of the visit (round-robin)
    n_res = max(min(total, 120), 4)   # exposures resident in HBM: one slot each (26 GB); a longer run cycles through them
    visit = synthetic.Visit(args.config, det, gr, cal, n_exposures=n_res * n_gpus)
    eng = engine.get_engine(device, gr, det, cal, visit.NSAMP, visit.SAMPSEQ, visit.SUBARRAY)
    ctx = eng.ctx

    def upload_all(stride, out_dtype, rng_mode):
        W = None
        for j in range(n_res):
            i = rank + j * n_gpus
            eg = ExposureGenerator(det, gr, visit.NSAMP, visit.SAMPSEQ, visit.SUBARRAY, calibration=cal,
                                   device=device, seed=visit.seed, exposure_index=i)
            desc = eg.build_descriptor(eng, out_dtype=out_dtype, rng_mode=rng_mode, **visit.frame_kwargs(i))
            ctx.upload(j * stride, desc)      # inputs resident in HBM before any timed region
            W = desc.n_wl
        ctx.synchronize()
        return W

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=grp)

    own_rates = []       # one entry per timed() call: what THIS rank did by its own clock (not the max over ranks)

    def timed(slot_of, steps, warmup, events_every=0):
        """Exactly `steps` exposures after `warmup` untimed ones, bracketed by barrier + synchronise;
        returns the elapsed seconds (max over ranks).  events_every = n: the selected kernels' HIP events are
        recorded on every n-th exposure only (an event pair costs the stream ~5 us either side of the kernel)."""
        elapsed = timed_pass(ctx, slot_of, steps, warmup, sync_all, events_every, device_sync=torch.cuda.synchronize)
        own_rates.append(steps / elapsed if (steps and elapsed > 0) else 0.0)    # this rank's own exposures/s
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
            elapsed = float(t.item())
            dist.barrier(group=grp)
        return elapsed

    out_dtype = np.float64 if args.out_f64 else np.float32
    rng_mode = _lib.RNG_SPLIT if args.thrower == "split" else _lib.RNG_PHILOX
    # slot j runs on stream j % 2: with --streams 1 only even slots are used
    stride = 1 if args.streams == 2 else 2
    W = upload_all(stride, out_dtype, rng_mode)

    def slot_of(j):
        return (j % n_res) * stride

    # timed region: HIP events around the roofline kernel only (every event pair costs a few microseconds
    # of stream time; the other kernels are timed in the breakdown pass below)
    timed(slot_of, 0, args.warmup)
    reruns_before = ctx.reruns
    ctx.profile_select(["k_ramp"])
    ctx.profile_enable(True)
    ctx.profile_reset()
    del own_rates[:]
    reps = [timed(slot_of, args.steps, 0, events_every=RAMP_EVENTS_EVERY) for _ in range(REPS)]
    rank_rates = list(own_rates)                      # this rank's exposures/s in each repetition of the timed region
    ctx.profile_enable(True)
    prof_ramp = ctx.profile_get()
    elapsed = float(np.median(reps))
    # every exposure of the timed region must have been complete: a slot whose status word says otherwise would have
    # been run a second time by a download -- which a loop of run() calls never makes
    # (every pass ends in ctx.synchronize(), which settles the slots -- wayne_ctx_synchronize -- so nothing can be left
    # incomplete; a second run it had to make is counted here, and its time is inside the repetition that made it)
    statuses = [ctx.status(slot_of(j)) for j in range(min(n_res, args.steps))]
    if any(statuses):
        raise SystemExit("rank %d: exposures of the timed region incomplete after synchronize (status %s)" % (
            rank, sorted(set(statuses))))
    reruns_timed = ctx.reruns - reruns_before

    ranks_reported = 1
    per_rank = {0: rank_rates}
    if dist is not None:
        got = [None] * world
        dist.all_gather_object(got, (rank, args.steps, rank_rates), group=grp)
        ranks_reported = len(set(r for r, _, _ in got))
        if ranks_reported != world or any(s != args.steps for _, s, _ in got):
            raise SystemExit("ranks disagree: %s" % (got,))
        per_rank = {r: rates for r, _, rates in got}

    # breakdown pass (not part of `value`): the same exposures again with every kernel timed
    ctx.profile_select(None)
    ctx.profile_reset()
    n_break = min(args.steps, 10)
    for j in range(args.warmup, args.warmup + n_break):
        ctx.run(slot_of(j))
    ctx.synchronize()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    prof["k_ramp"] = {"launches": prof_ramp["k_ramp"]["launches"], "ms": prof_ramp["k_ramp"]["ms"]}   # timed region

    # sustained rate (not `value`): the same exposures back to back on the same stream for >= 2.5 s, cycling through
    # the resident slots -- seconds of VALU-saturated kernels under the power cap instead of a 10 ms burst
    sustained = None
    if not args.no_extra_pass:
        n_sus = int(max(args.steps, np.ceil(SUSTAIN_S * args.steps / elapsed)))
        e_sus = timed(slot_of, n_sus, 0)
        sustained = {"value": n_sus * n_gpus / e_sus, "unit": "exposures/s", "steps": n_sus, "seconds": e_sus,
                     "vs_value": (n_sus / e_sus) / (args.steps / elapsed),
                     "note": "%d exposures per rank back to back on one stream (the timed region's workload and slots), "
                             "barrier + synchronise either side, max over ranks" % n_sus}

    # sanity: the last exposure really produced a frame
    reads = ctx.download(slot_of(args.steps - 1))
    assert np.isfinite(reads).all() and reads[-1].max() > 100.0

    extras = {}
    out_mb = (eng.R + 1) * eng.S * eng.S * 4 / 1e6
    n_x = min(args.steps, 30)

    def median_rate(sl, steps, reps=3, warmup=None):
        """Median of `reps` passes of exactly `steps` exposures (each its own barrier + synchronise bracket); the
        first pass is preceded by `warmup` untimed exposures.  (median_of_passes with the ranks' max-reduce;
        scripts/bench_configs.py calls median_of_passes for its one-stream / two-stream columns: same warm-up, same
        step count, same passes -- the two scripts cannot disagree by construction.)"""
        w = args.warmup if warmup is None else warmup
        vals = []
        for rep in range(reps):
            vals.append(steps * n_gpus / timed(sl, steps, w if rep == 0 else 0))
        return float(np.median(vals)), vals

    def ramp_events_pass(sl, steps):
        """`steps` exposures with k_ramp's own HIP events on every RAMP_EVENTS_EVERY-th -> (exposures/s, ms per launch)."""
        ctx.profile_select(["k_ramp"])
        ctx.profile_enable(True)
        ctx.profile_reset()
        e = timed(sl, steps, args.warmup, events_every=RAMP_EVENTS_EVERY)
        ctx.profile_enable(True)
        pr = ctx.profile_get()["k_ramp"]
        ctx.profile_enable(False)
        ctx.profile_select(None)
        return steps * n_gpus / e, pr["ms"] / max(pr["launches"], 1), pr["launches"]

    ramp_variant = ctx.ramp_variant(slot_of(0))          # the instantiation `value` and `roofline` were measured on
    use_box0, _, segs0 = ctx.debug_boxes(slot_of(0))

    if n_gpus == 1 and not args.no_extra_pass:
        # (0) k_ramp with its calibration planes EVICTED between launches.  Every launch re-reads the same 159 MB of
        # dark / pixel-flat / sky / linearity planes, which fit the 256 MB Infinity Cache, and FETCH_SIZE counts hits
        # there (MI355X_MICROARCH.md:297-309): the warm figure above may be a fraction of a more generous ceiling than
        # HBM's.  Here a 1 GiB fill runs on the stream between an exposure's thrower and its k_ramp (outside the
        # kernel's own start / stop events): same kernel, same bytes, cold planes.
        try:
            ext = torch.cuda.ExternalStream(ctx.stream)        # streams[0]: the stream of every even slot
            scrub = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:%d" % device)
            ctx.profile_select(["k_ramp"])
            ctx.profile_enable(True)
            ctx.profile_reset()
            n_cold = min(args.steps, 24)
            # (even slots only: they run on streams[0], the stream the fill is enqueued on -- with --streams 2 the odd
            # slots' k_ramp would be neither ordered after the fill nor evicted)
            n_even = n_res if stride == 2 else max(n_res // 2, 1)

            def cold_slot(j):
                return (j % n_even) * 2
            for j in range(n_cold):
                ctx.run_front(cold_slot(j))
                with torch.cuda.stream(ext):
                    scrub.fill_(j & 0xFF)
                ctx.run_back(cold_slot(j))
            ctx.synchronize()
            torch.cuda.synchronize()
            pr = ctx.profile_get()["k_ramp"]
            ctx.profile_enable(False)
            ctx.profile_select(None)
            del scrub
            extras["ramp_cold"] = {"ms_per_launch": pr["ms"] / max(pr["launches"], 1), "launches_timed": pr["launches"],
                                   "evicted_by": "1 GiB device fill on the kernel's stream before every launch"}
        except Exception as e:                               # (a measurement, not the product: say so and go on)
            extras["ramp_cold"] = {"error": repr(e)}
        # (1) exposures alternating over the context's two HIP streams (prep / ramp of one under the thrower of the next)
        if args.streams == 1:
            upload_all(1, out_dtype, rng_mode)
            n_two = max(args.steps, 40)
            med, vals = median_rate(lambda j: j % n_res, n_two, warmup=4)
            extras["two_streams"] = {"value": med, "unit": "exposures/s", "repetitions": vals, "steps_each": n_two,
                                     "note": "what VisitRunner (Observation, the CLI) delivers device-side: the "
                                             "same exposures alternating over the context's two HIP streams; median of 3 passes of %d "
                                             "after 4 warm-up exposures (bench.timed_pass: the function "
                                             "scripts/bench_configs.py uses too)" % n_two}
        # (2) float64 reads, the reference's SCI dtype (exposure.py:133-214) -- a different k_ramp instantiation (fp64
        # cumulative sum, 8-byte stores) with a roofline block of its own
        if not args.out_f64:
            upload_all(2, np.float64, rng_mode)
            v64, ms64, n64 = ramp_events_pass(slot_of, n_x)
            rb64 = ramp_bytes(eng.N, eng.S, eng.R, 8, int(segs0.sum()) if use_box0 else None)
            extras["out_f64"] = {"value": v64, "unit": "exposures/s",
                                 "note": "float64 reads (%.0f MB written per exposure instead of %.0f), one stream" % (
                                     2 * out_mb, out_mb),
                                 "roofline": {"bound": "hbm", "kernel": ctx.ramp_variant(slot_of(0)),
                                              "achieved": rb64 / (ms64 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": rb64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                              "bytes_per_launch": rb64, "ms_per_launch": ms64, "launches_timed": n64}}
        # (3) every electron thrown one by one, as the reference does (pyparallel_menu.c:87-108)
        if args.thrower == "split":
            upload_all(2, np.float32, _lib.RNG_PHILOX)
            extras["per_electron"] = {"value": n_x / timed(slot_of, n_x, args.warmup), "unit": "exposures/s",
                                      "note": "rng_mode PHILOX: all %.3g electrons thrown individually, f32 reads, "
                                              "one stream" % (prof["electrons"] / max(n_break, 1))}
            upload_all(2, np.float64, _lib.RNG_PHILOX)
            extras["per_electron_f64"] = {"value": n_x / timed(slot_of, n_x, args.warmup), "unit": "exposures/s",
                                          "note": "every electron thrown AND float64 reads: the reference's arithmetic "
                                                  "shape, one stream"}
            # (3b) the bit-exact mode: glibc rand_r streams + the reference's OpenMP partition replayed on the device,
            # fp64 Box-Muller and positions -- the thrower whose frames equal the reference C's bit for bit
            # (tests/test_psf_gpu.py, tests/golden/psf_*.npz) -- with float64 reads
            upload_all(2, np.float64, _lib.RNG_REPLAY)
            n_rep = min(n_x, 10)
            extras["replay_bit_exact"] = {"value": n_rep / timed(slot_of, n_rep, args.warmup), "unit": "exposures/s",
                                          "note": "rng_mode REPLAY (threads_compat 2): the thrower that reproduces the "
                                                  "reference's frames bit for bit, float64 reads, one stream"}

    if not args.no_extra_pass:
        # (4) + (5): SURVEY 8(d)'s second leg, at EVERY N -- each rank drains its own GPU through its own pinned buffers
        # and PCIe root; a pass is bracketed by barrier + synchronise and timed by the slowest rank, every rank's own
        # rate is listed beside it (the host side -- NUMA, thread pools, eight ranks' copies -- is the only place this
        # path can fail to scale: observation.py:396-413 is the reference's axis)
        n_d = max(2 * n_x, 40)

        def host_leg(fn, reps=3):
            vals, own = [], []
            for _ in range(reps):
                sync_all()
                t = time.perf_counter()
                fn()
                dt_own = time.perf_counter() - t
                own.append(n_d / dt_own)
                dt_max = dt_own
                if dist is not None:
                    tt = torch.tensor([dt_own], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=grp)
                    dt_max = float(tt.item())
                vals.append(n_d * n_gpus / dt_max)
            per = {0: own}
            if dist is not None:
                got_ = [None] * world
                dist.all_gather_object(got_, (rank, own), group=grp)
                per = {r_: o_ for r_, o_ in got_}
            med = float(np.median(vals))
            return {"value": med, "unit": "exposures/s", "GB_per_s": med * out_mb / 1e3,
                    "frac_of_pcie": med * out_mb / 1e3 / (PCIE_GBS * n_gpus), "repetitions": vals,
                    "ranks_reported": len(per),
                    "per_rank_exposures_s": {str(r_): [round(x, 1) for x in per[r_]] for r_ in sorted(per)}}

        # delivered: reads of resident exposures copied to pinned host memory through the VisitRunner pipeline
        # (4 slots in rotation over both streams, device-to-host copies on the slots' own streams)
        runner = wvisit.VisitRunner(visit, device=device, out_dtype=np.float32)
        upload_all(1, np.float32, _lib.RNG_SPLIT)
        runner.run_resident(8)                       # warm-up: pinned buffers are allocated on first use
        extras["delivered"] = host_leg(lambda: runner.run_resident(n_d))
        extras["delivered"]["note"] = ("device-resident descriptors; reads copied to pinned host memory (PCIe-inclusive; "
                                       "%.1f MB per exposure), VisitRunner pipeline on every rank, median of 3 passes of %d "
                                       "per rank, each timed by the slowest rank" % (out_mb, n_d))
        # end to end: descriptor build + upload + kernels + fetch per exposure, light curves on the device
        runner_lc = wvisit.VisitRunner(visit, device=device, out_dtype=np.float32, device_lc=True)
        mine = [(rank + j * n_gpus) % visit.n_exposures for j in range(n_d)]          # this rank's round-robin share
        runner_lc.run(mine[:8])
        extras["end_to_end"] = host_leg(lambda: runner_lc.run(mine))
        extras["end_to_end"]["note"] = ("per exposure: host descriptor (K-vectors) -> upload -> k_lightcurve + all kernels -> "
                                        "reads in pinned host memory; VisitRunner, device light curves (no K x W upload), "
                                        "median of 3 passes of %d per rank, each timed by the slowest rank" % n_d)

    if n_gpus == 1 and not args.no_extra_pass:
        # (6) the inner drop-in boundary by itself: wayne_psf_apply (= pyparallel.apply_psf, pyparallel.pyx:14-38) for one
        # sub-sample of this workload -- host arrays in, the frame out, replay mode (the reference's frame bit for bit)
        if rank == 0:
            rng = np.random.default_rng(3)
            shape = visit.stellar_flux / visit.stellar_flux.sum()
            counts = rng.poisson(shape * visit.E / visit.K).astype(np.int32)
            xs = np.linspace(560.0, 700.0, counts.size)
            ys = 300.0 + 0.01 * (xs - 560.0)
            polys = [np.polyval(getattr(visit.grism, n_), visit.wl) for n_ in ("psf_ratio_poly", "psf_sigmal_poly", "psf_sigmah_poly")]
            side = eng.N
            for j in range(3):
                ctx.psf_apply(counts, xs, ys, polys[0], polys[1], polys[2], side, side, 7 + j, 4, _lib.RNG_REPLAY)
            n_calls = 30
            t = time.perf_counter()
            for j in range(n_calls):
                ctx.psf_apply(counts, xs, ys, polys[0], polys[1], polys[2], side, side, 7 + j, 4, _lib.RNG_REPLAY)
            dt = (time.perf_counter() - t) / n_calls
            extras["psf_apply_replay"] = {"ms_per_call": dt * 1e3, "electrons_per_s": float(counts.sum()) / dt,
                                          "electrons_per_call": int(counts.sum()),
                                          "note": "wayne_psf_apply through the C ABI, rng_mode REPLAY, threads 4: one sub-sample "
                                                  "of the workload, host arrays in and the %d x %d frame out (PCIe-inclusive)" % (side, side)}

    if rank == 0:
        N, S, R, K = eng.N, eng.S, eng.R, visit.K
        ob = 8 if args.out_f64 else 4
        launches = max(prof["k_ramp"]["launches"], 1)
        ramp_ms = prof["k_ramp"]["ms"] / launches
        use_box, segs = use_box0, segs0          # (of the timed region's slots: later passes re-upload them)
        rb = ramp_bytes(N, S, R, ob, int(segs.sum()) if use_box else None)
        rb_all = ramp_bytes(N, S, R, ob)
        achieved = rb / (ramp_ms * 1e-3) / 1e9
        sb = survey_bytes(N, S, R, K, W, ob, 1.64e5)
        throw_ms = prof["k_throw"]["ms"] / max(prof["k_throw"]["launches"], 1)
        narrow_ms = prof["k_narrow"]["ms"] / max(prof["k_narrow"]["launches"], 1)
        # knob fork_narrow (WAYNE_FORK_NARROW=1 at context creation; default off): the library launches k_narrow on a side stream beside k_lane and
        # the k_throw profile interval then covers both kernels
        forked = args.thrower == "split" and bool(ctx.get_knob("fork_narrow"))
        # (the k_throw interval always includes k_lane, which follows it on the slot's stream)
        thrower_ms = throw_ms if forked else throw_ms + narrow_ms
        electrons = prof["electrons"] / max(n_break, 1)
        rates = sorted(args.steps * n_gpus / e for e in reps)
        line = {
            "metric": "simulated WFC3-IR exposures/sec (1014x1014, NSAMP=16, spatial scan)",
            "value": args.steps * n_gpus / elapsed, "unit": "exposures/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_label(ramp_variant, args.out_f64),
            "data": "synthetic",
            "ranks_reported": ranks_reported,
            "repetitions": {"n": REPS, "steps_each": args.steps, "values": [args.steps * n_gpus / e for e in reps],
                            "median": rates[len(rates) // 2], "min": rates[0], "max": rates[-1],
                            "per_rank_exposures_s": {str(r): [round(x, 1) for x in per_rank[r]] for r in sorted(per_rank)},
                            "incomplete_exposures": 0, "second_runs_in_timed_region": reruns_timed,
                            "note": "value = the median repetition; each is exactly `steps` exposures per rank between "
                                    "barrier + synchronise, max over ranks; per_rank_exposures_s: every rank's own rate "
                                    "in each repetition by its own clock (an imbalance shows here, not in the max)"},
            "config": {"workload": "%s: %s spatial scan %g px/s, SUBARRAY=%d (frame %dx%d), %s NSAMP=%d, "
                                   "K=%d sub-samples, W=%d bins, %.3g electrons/exposure, all detector effects on "
                                   "(flat, sky, cosmic rays, gain, dark, non-linearity, clip, read noise), "
                                   "thrower=%s, %s, %s, device-complete reads in HBM" % (
                                       args.config, gr.name, visit.scan_speed, visit.SUBARRAY, N, N, visit.SAMPSEQ,
                                       visit.NSAMP, K, W, electrons,
                                       "split (wide component per electron, narrow component multinomial)"
                                       if args.thrower == "split" else "per-electron",
                                       READS_LABEL[bool(args.out_f64)],
                                       "one HIP stream" if args.streams == 1 else "two HIP streams (what VisitRunner runs)"),
                       "exposures_per_rank": args.steps, "sharding": "round-robin exposures, no collective"},
            "roofline": {"bound": "hbm", "kernel": ramp_variant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_launch": rb, "ms_per_launch": ramp_ms,
                         "accumulator_segments_loaded": int(segs.sum()) if use_box else None,
                         "bytes_per_launch_loading_every_accumulator": rb_all,
                         "achieved_loading_every_accumulator": rb_all / (ramp_ms * 1e-3) / 1e9,
                         "note": "frac = algorithmic bytes / the kernel's own duration in the TIMED REGION, i.e. in the "
                                 "steady state of whole exposures on one stream: the 159 MB of calibration planes every "
                                 "launch re-reads (dark SCI / ERR, pixel flat, sky, linearity) may then be served by the "
                                 "256 MB Infinity Cache rather than HBM -- `cold_cache` is the same kernel timed with those "
                                 "planes evicted before every launch (1 GiB fill on its stream), same bytes: the "
                                 "fraction of the 8 TB/s roof that is certainly HBM.  bytes_per_launch counts the int64 "
                                 "accumulators only where the kernel loads them (the per-read boxes of the thrower's "
                                 "reach); the r01/r02 kernel loaded all of them: that byte count and the GB/s it would "
                                 "give are listed beside it, not used for `frac`",
                         "launches_timed": prof_ramp["k_ramp"]["launches"],
                         "timing": "HIP events on the kernel's own stream, every %d-th launch of the timed region" % RAMP_EVENTS_EVERY,
                         "survey_formula_bytes_per_exposure": sb,
                         "achieved_survey_formula": sb / (ramp_ms * 1e-3) / 1e9},
            "kernels_ms_per_exposure": {k: v["ms"] / max(v["launches"], 1) for k, v in prof.items() if k != "electrons"},
            "thrower": {"mode": args.thrower, "electrons_per_exposure": electrons, "ms": thrower_ms,
                        "electrons_per_s": electrons / (thrower_ms * 1e-3) if thrower_ms > 0 else None,
                        "note": "split mode: the k_throw interval spans k_throw (bins beyond a lane's cap: normally none) + "
                                "k_lane on the slot's stream with k_narrow beside them on a side stream"
                        if forked else "k_throw + k_lane (one interval), then k_narrow, on one stream"},
        }
        lib_path, lib_flags = _lib.library_info()
        line["library"] = {"path": os.path.relpath(lib_path, ROOT), "build_flags": lib_flags, "abi": _lib.ABI_VERSION}
        if lib_flags:
            # a negative-control or timing build (its sources say "wrong frames"): not a measurement of the product
            line["value"] = None
            line["library"]["note"] = "value withheld: this library was built with %s" % lib_flags
        line["sustained"] = sustained
        cold = extras.pop("ramp_cold", None)
        if cold and "ms_per_launch" in cold and cold["ms_per_launch"] > 0:
            cold["achieved"] = rb / (cold["ms_per_launch"] * 1e-3) / 1e9
            cold["frac"] = cold["achieved"] / HBM_PEAK_GBS
            cold["warm_over_cold"] = ramp_ms / cold["ms_per_launch"]
        line["roofline"]["cold_cache"] = cold
        line.update(extras)
        # numbers that only a rocprofv3 --pmc run can give come from profiles/*.json, which
        # scripts/collect_profiles.sh stamps with the hash of wayne_amd/csrc they were measured on: quoted only
        # while that is the code in the tree
        here = csrc_hash()
        key = "%s/%s" % (args.config, "f64" if args.out_f64 else "f32")
        tf = os.path.join(ROOT, "profiles", "k_ramp_traffic.json")
        t = json.load(open(tf)) if os.path.exists(tf) else {}
        if t.get(key) and t.get("csrc_hash") == here:
            line["roofline"]["traffic"] = t[key]["hbm_bytes_per_launch"]
            line["roofline"]["traffic_source"] = t[key]["source"]
        else:
            line["roofline"]["traffic_source"] = "null: " + (
                "profiles/k_ramp_traffic.json was measured on csrc %s, the tree is %s" % (t.get("csrc_hash"), here)
                if t else "no profiles/k_ramp_traffic.json")
        vf = os.path.join(ROOT, "profiles", "valu_issue.json")
        v = json.load(open(vf)) if os.path.exists(vf) else {}
        if args.config == "cfg4" and args.thrower == "split":
            if v.get("csrc_hash") == here:
                # counted and un-clamped (scripts/make_profile_stamps.py): quad-cycles of vector execution over SIMD cycles
                line["thrower"]["valu_issue"] = {k: {"valu_busy_frac": x["valu_busy_frac"],
                                                     "valu_busy_frac_gui": x.get("valu_busy_frac_gui"),
                                                     "lane_utilisation": x["lane_utilisation"]}
                                                 for k, x in v["kernels"].items()}
                line["thrower"]["valu_issue_note"] = "valu_busy_frac: vector quad-cycles per SIMD / (kernel duration x 2.4 GHz): a lower bound, un-clamped"
                line["thrower"]["valu_issue_source"] = v.get("source", "profiles/valu_issue.json")
            else:
                line["thrower"]["valu_issue"] = None
                line["thrower"]["valu_issue_source"] = "null: profiles/valu_issue.json was measured on csrc %s, the tree is %s" % (
                    v.get("csrc_hash"), here)
        if not args.no_cpu_baseline and n_gpus == 1:
            line["cpu_baseline"] = cpu_baseline(visit)
        else:
            line["cpu_baseline"] = None
        emit_line(line)
    if dist is not None:
        dist.barrier(group=grp)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
