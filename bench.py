#!/usr/bin/env python3
"""bench.py -- simulated WFC3-IR exposures/sec on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4]

One "step" = one whole exposure (1014x1014 frame, NSAMP = 16, 128-sub-sample
spatial scan, 1e9 electrons: BASELINE.json configs[3], the configuration the
metric is quoted on) synthesised by the HIP path: k_prep_wl, k_prep_sub,
k_throw, k_cosmic, k_ramp.  All inputs and calibration planes are resident in
HBM before the timed region; outputs stay in HBM (device-complete rate).
Exposures are independent: with N ranks each rank runs its own K exposures
(round-robin exposure indices, no collective in the data path) -> weak scaling.

Prints ONE JSON line on rank 0 (the driver's contract) with two extra objects:
  roofline      the fused up-the-ramp kernel k_ramp against the HBM roof
  cpu_baseline  the reference's C thrower + the numpy restatement of the host
                loop, timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def ramp_bytes(N, S, R, out_bytes):
    """Compulsory HBM bytes of ONE k_ramp launch as designed (DESIGN.md "k_ramp"):
    per interior pixel and read: int64 accumulator read + cleared (8 + 8), dark SCI + ERR
    (4 + 4), read written (out_bytes); once per pixel: pixel flat 4, sky 4, c1..c4 16,
    zero read written (out_bytes).  Border pixels only write their reads."""
    inner = N * N
    per_read_inner = 8 + 8 + 4 + 4
    once_inner = 4 + 4 + 16
    return R * (inner * per_read_inner + S * S * out_bytes) + inner * once_inner + S * S * out_bytes


def survey_bytes(N, S, R, K, W, out_bytes, A_fp):
    """R (12 N^2 + 28 S^2 + B_out S^2) + (4 + B_out) S^2 + 8 K W + 16 A_fp
    -- the unfused reference structure (gain / sky / linearity / zero read re-read per read)."""
    return R * (12 * N * N + 28 * S * S + out_bytes * S * S) + (4 + out_bytes) * S * S + 8 * K * W + 16 * A_fp


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
    N = 1014 if visit.SUBARRAY == 1024 else visit.SUBARRAY
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

    # choose the faster OpenMP team size on this box (the reference's default yml uses 4)
    ncpu = os.cpu_count() or 1
    best = None
    for th in sorted(set([1, min(4, ncpu)])):
        threads_used[0] = th
        dt, ne, _ = one_subsample(0)
        if best is None or dt < best[0]:
            best = (dt, th)
    threads_used[0] = best[1]
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
    return {"value": 1.0 / per_exposure, "unit": "exposures/s", "cores": int(threads_used[0]), "kind": kind,
            "sample": "%s exposure 0: %d of %d sub-samples (%.3g electrons, %.2f s each: C thrower threads=%d + "
                      "numpy passes), 1 of %d per-read stages (%.2f s), post-ramp stage on 2 reads (%.2f s/read); "
                      "extrapolated to one exposure = %.1f s" % (visit.name, n_sub, visit.K, electrons,
                                                                 float(np.mean(t_sub)), threads_used[0], R, t_read,
                                                                 t_post, per_exposure),
            "host_cpus": ncpu, "thrower_electrons_per_s": electrons / float(np.sum(t_sub))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out-f64", action="store_true", help="float64 reads (the reference's dtype)")
    ap.add_argument("--thrower", default="split", choices=["split", "electron"],
                    help="split: narrow PSF component drawn as a multinomial (WAYNE_RNG_SPLIT, same distribution); "
                         "electron: every electron thrown individually (WAYNE_RNG_PHILOX)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="HIP streams in the timed region (1: kernels never co-run, so per-kernel event times are "
                         "clean; the 2-stream rate is reported separately as two_streams)")
    ap.add_argument("--no-extra-pass", action="store_true",
                    help="skip the additional two-stream pass (use under rocprofv3 so that the kernel statistics "
                         "cover the timed region only)")
    args = ap.parse_args()
    os.environ["WAYNE_STREAMS"] = "2"        # the context always owns two streams; slots select them

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    n_gpus = max(world, 1)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    # one process per GPU; WAYNE_BENCH_SHARE_GPU=1 lets several ranks share device 0 (a rehearsal of the
    # multi-process path on a one-GPU box, with the gloo backend since RCCL refuses duplicate devices)
    share = os.environ.get("WAYNE_BENCH_SHARE_GPU") == "1"
    device = 0 if share else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            # "nccl" is RCCL on ROCm; used for the barrier and the max-over-ranks of the elapsed time only
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device))

    from wayne_amd import _lib, calibration, detector, engine, grism, synthetic

    cal = calibration.CalibrationSet.synthetic(11)
    det = detector.WFC3_IR()
    cfg = synthetic.CONFIGS[args.config]
    gr = grism.G141(cal) if cfg["grism"] == "G141" else grism.G102(cal)
    total = args.warmup + args.steps
    # exposure j of this rank is exposure index rank + j * n_gpus of the visit (round-robin)
    visit = synthetic.Visit(args.config, det, gr, cal, n_exposures=min(total, 120) * n_gpus)
    eng = engine.get_engine(device, gr, det, cal, visit.NSAMP, visit.SAMPSEQ, visit.SUBARRAY)
    ctx = eng.ctx
    # exposures resident in HBM: one slot each, at most 120 (26 GB); a longer run cycles through them again
    n_res = min(total, 120)

    from wayne_amd.exposure_generator import ExposureGenerator
    out_dtype = np.float64 if args.out_f64 else np.float32
    rng_mode = _lib.RNG_SPLIT if args.thrower == "split" else _lib.RNG_PHILOX
    W = None
    for j in range(n_res):
        i = rank + j * n_gpus
        eg = ExposureGenerator(det, gr, visit.NSAMP, visit.SAMPSEQ, visit.SUBARRAY, calibration=cal,
                               device=device, seed=visit.seed, exposure_index=i)
        desc = eg.build_descriptor(eng, out_dtype=out_dtype, rng_mode=rng_mode, **visit.frame_kwargs(i))
        ctx.upload(j * (1 if args.streams == 2 else 2), desc)      # inputs resident in HBM before the timed region
        W = desc.n_wl
    ctx.synchronize()

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # slot j runs on stream j % 2: with --streams 1 only even slots are used
    stride = 1 if args.streams == 2 else 2

    def slot_of(j):
        return (j % n_res) * stride

    for j in range(args.warmup):
        ctx.run(slot_of(j))
    sync_all()
    # timed region: HIP events around the roofline kernel only (every event pair costs a few microseconds
    # of stream time; the other kernels are timed in the breakdown pass below)
    ctx.profile_select(["k_ramp"])
    ctx.profile_enable(True)
    ctx.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for j in range(args.warmup, total):
        ctx.run(slot_of(j))
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    prof_ramp = ctx.profile_get()
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

    # extra pass: the same exposures alternating over the context's two HIP streams
    # (prep / ramp of one exposure co-run with the thrower of another)
    two = None
    if args.streams == 1 and n_gpus == 1 and not args.no_extra_pass:
        for j in range(n_res):
            ctx.upload(j, ExposureGenerator(det, gr, visit.NSAMP, visit.SAMPSEQ, visit.SUBARRAY, calibration=cal,
                                            device=device, seed=visit.seed, exposure_index=rank + j * n_gpus
                                            ).build_descriptor(eng, out_dtype=out_dtype, rng_mode=rng_mode,
                                                               **visit.frame_kwargs(rank + j * n_gpus)))
        for j in range(args.warmup):
            ctx.run(j % n_res)
        sync_all()
        t1 = time.perf_counter()
        for j in range(args.warmup, total):
            ctx.run(j % n_res)
        ctx.synchronize()
        torch.cuda.synchronize()
        two = args.steps / (time.perf_counter() - t1)
        stride = 1

    # delivered: the same exposures with their reads copied into pinned host memory (copy of exposure n
    # overlapping the kernels of n + 1), as a visit driver consumes them
    delivered = None
    if two is not None and n_res >= 4:
        pending = []
        ring = 4                             # exposures in flight: their pinned buffers are allocated in the warm-up
        for j in range(max(args.warmup, ring)):
            ctx.run(j % ring)
            ctx.fetch_async(j % ring)
            ctx.wait(j % ring)
        sync_all()
        t2 = time.perf_counter()
        for j in range(args.warmup, total):
            slot = j % ring
            ctx.run(slot)
            ctx.fetch_async(slot)
            pending.append(slot)
            if len(pending) > 2:
                ctx.wait(pending.pop(0))
        while pending:
            ctx.wait(pending.pop(0))
        delivered = args.steps / (time.perf_counter() - t2)

    # sanity: the last exposure really produced a frame
    reads = ctx.download(slot_of(total - 1) if two is None else (total - 1) % n_res)
    assert np.isfinite(reads).all() and reads[-1].max() > 100.0

    if rank == 0:
        N, S, R, K = eng.N, eng.S, eng.R, visit.K
        ob = 8 if args.out_f64 else 4
        launches = max(prof["k_ramp"]["launches"], 1)
        ramp_ms = prof["k_ramp"]["ms"] / launches
        rb = ramp_bytes(N, S, R, ob)
        achieved = rb / (ramp_ms * 1e-3) / 1e9
        sb = survey_bytes(N, S, R, K, W, ob, 1.64e5)
        throw_ms = prof["k_throw"]["ms"] / max(prof["k_throw"]["launches"], 1)
        narrow_ms = prof["k_narrow"]["ms"] / max(prof["k_narrow"]["launches"], 1)
        # WAYNE_FORK_NARROW (default on): the library launches k_narrow on a side stream beside k_throw and
        # its k_throw profile interval then covers both kernels
        forked = args.thrower == "split" and os.environ.get("WAYNE_FORK_NARROW", "1") != "0"
        thrower_ms = throw_ms if forked else throw_ms + narrow_ms
        electrons = prof["electrons"] / max(n_break, 1)
        line = {
            "metric": "simulated WFC3-IR exposures/sec (1014x1014, NSAMP=16, spatial scan)",
            "value": args.steps * n_gpus / elapsed, "unit": "exposures/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 ramp arithmetic, f32 thrower, int32/int64 accumulation; %s reads" % (
                "f64" if args.out_f64 else "f32"),
            "data": "synthetic",
            "config": {"workload": "%s: %s spatial scan %g px/s, SUBARRAY=%d (frame %dx%d), %s NSAMP=%d, "
                                   "K=%d sub-samples, W=%d bins, %.3g electrons/exposure, all detector effects on "
                                   "(flat, sky, cosmic rays, gain, dark, non-linearity, clip, read noise), "
                                   "thrower=%s, device-complete reads in HBM" % (
                                       args.config, gr.name, visit.scan_speed, visit.SUBARRAY, N, N, visit.SAMPSEQ,
                                       visit.NSAMP, K, W, electrons,
                                       "split (wide component per electron, narrow component multinomial)"
                                       if args.thrower == "split" else "per-electron"),
                       "exposures_per_rank": args.steps, "sharding": "round-robin exposures, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_ramp", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_launch": rb, "ms_per_launch": ramp_ms,
                         "survey_formula_bytes_per_exposure": sb,
                         "achieved_survey_formula": sb / (ramp_ms * 1e-3) / 1e9},
            "kernels_ms_per_exposure": {k: v["ms"] / max(v["launches"], 1) for k, v in prof.items() if k != "electrons"},
            "thrower": {"mode": args.thrower, "electrons_per_exposure": electrons, "ms": thrower_ms,
                        "electrons_per_s": electrons / (thrower_ms * 1e-3) if thrower_ms > 0 else None,
                        "note": "k_narrow runs beside k_throw on a side stream: the k_throw interval spans both"
                        if forked else "k_throw then k_narrow on one stream"},
            "two_streams": None if two is None else {"value": two, "unit": "exposures/s",
                                                     "note": "same exposures alternating over two HIP streams"},
            "delivered": None if delivered is None else {
                "value": delivered, "unit": "exposures/s",
                "note": "reads copied to pinned host memory (PCIe-inclusive; %.1f MB per exposure)" % (
                    (eng.R + 1) * eng.S * eng.S * (8 if args.out_f64 else 4) / 1e6)},
        }
        traffic_file = os.path.join(ROOT, "profiles", "k_ramp_traffic.json")
        if os.path.exists(traffic_file):
            t = json.load(open(traffic_file)).get("%s/%s" % (args.config, "f64" if args.out_f64 else "f32"))
            if t:
                line["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = t["source"]
        issue_file = os.path.join(ROOT, "profiles", "valu_issue.json")
        if os.path.exists(issue_file) and args.config == "cfg4" and args.thrower == "split":
            # the thrower kernels are bound by VALU issue, not by HBM: committed PMC summary of this workload
            v = json.load(open(issue_file))
            line["thrower"]["valu_issue"] = {k: {"frac": x["valu_issue_frac"], "lane_utilisation": x["lane_utilisation"]}
                                             for k, x in v["kernels"].items()}
            line["thrower"]["valu_issue_source"] = "profiles/valu_issue.json (rocprofv3 --pmc SQ counters, profiles/r01/final_pmc_sq.json)"
        if not args.no_cpu_baseline and n_gpus == 1:
            line["cpu_baseline"] = cpu_baseline(visit)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
