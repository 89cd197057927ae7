"""Command line: generate a visit from a YAML parameter file.

    python -m wayne_amd.run_visit -p <parameter_file> [--calibration DIR] [--device N] [--max-exposures M] [--gpus G] [--resume]

Accepts the reference's parameter files (wayne/run_visit.py:1-9, example
examples/hd209458b_12181_simulation_parameters.yml): sections `general`
(outdir, seed, threads), `target`, `observation`, `trends`.  Differences:

  * calibration files come from --calibration DIR (the reference downloads
    them at import, params.py:41-56); without it seeded synthetic planes are used;
  * planet parameters come from the YAML (`period`, `sma`, `stellar_radius`,
    `inclination`, `eccentricity`, `periastron`, `transit_time`, `ldcoeffs`);
    the Open Exoplanet Catalogue lookup (oec.py) is not provided;
  * a missing stellar spectrum file falls back to a black body of
    `target: star_temperature` (default 6100 K);
  * `--resume`: exposures whose files are already in the output directory (whole, and this visit's: same start time and
    mode) are skipped; files are written under a temporary name and renamed, so an interrupted run leaves no partial
    file under a final name.  A resumed visit's files are those of an uninterrupted one (per-exposure Philox keys);
  * `--gpus G`: the process starts G rank processes itself (one per GPU of this node, before anything touches a
    GPU) and waits for them; under an external launcher (WORLD_SIZE / RANK set, one process per GPU) it is one
    rank.  Each rank generates its round-robin share of the exposures (observation.py:403-405 is the axis) on the
    CPUs of its GPU's NUMA node.  `--ranks-per-gpu R` starts G x R ranks, R to a GPU (rank r on device r // R):
    on small sub-arrays a visit is bound by one interpreter's lock, not by the GPU.
"""
import argparse
import os
import shutil
import sys

import numpy as np
import yaml

from . import calibration as _cal
from . import detector, grism, launch, observation, tools
from .trend_generators import scan_speed_varations


class WFC3SimConfigError(Exception):
    pass


def _get(d, key, default=None):
    try:
        v = d[key]
    except (KeyError, TypeError):
        return default
    return v


def build_observation(cfg, base_dir=".", calibration=None, device=0):
    """YAML dict -> configured Observation (the body of run_visit.run, run_visit.py:41-314)."""
    general, target, obs_cfg = cfg["general"], cfg["target"], cfg["observation"]
    outdir = general["outdir"]
    seed = _get(general, "seed") or 0
    cal = calibration if calibration is not None else _cal.CalibrationSet.synthetic(seed)
    grisms = {"G141": grism.G141, "G102": grism.G102}
    chosen_grism = grisms[obs_cfg["grism"]](cal)
    det = detector.WFC3_IR()

    def path(p):
        return p if os.path.isabs(p) else os.path.join(base_dir, p)

    rebin_resolution = _get(target, "rebin_resolution")
    planet_spectrum_file = _get(target, "planet_spectrum_file")
    transmission = bool(planet_spectrum_file)
    depth_planet = wl_planet = None
    # `ra` / `dec` (degrees, J2000) are an extension of the reference's schema: it reads them off the Open
    # Exoplanet Catalogue entry of `name`; with them the light-curve times become heliocentric (observation.py:340)
    planet = observation.Planet(name=str(_get(target, "name", "planet")),
                                star_temperature=_get(target, "star_temperature", 6100.0),
                                ra_deg=_get(target, "ra", None), dec_deg=_get(target, "dec", None))
    if transmission:
        wl_planet, depth_planet = tools.load_and_sort_spectrum(path(planet_spectrum_file))
        wl_planet, depth_planet = tools.crop_spectrum(0.9, 1.8, wl_planet, depth_planet)      # run_visit.py:152-153
        if rebin_resolution:
            new_wl = tools.wl_at_resolution(rebin_resolution, chosen_grism.wl_limits[0], chosen_grism.wl_limits[1])
            depth_planet = tools.rebin_spec(wl_planet, depth_planet, new_wl)
            wl_planet = new_wl
    stellar_file = _get(target, "stellar_spectrum_file")
    if stellar_file and os.path.exists(path(stellar_file)):
        wl_star, flux_star = tools.load_pheonix_stellar_grid_fits(path(stellar_file))
        if transmission:
            flux_star = tools.rebin_spec(wl_star, flux_star, wl_planet)
        elif rebin_resolution:
            new_wl = tools.wl_at_resolution(rebin_resolution, chosen_grism.wl_limits[0], chosen_grism.wl_limits[1])
            flux_star = tools.rebin_spec(wl_star, flux_star, new_wl)
            wl_star = new_wl
    elif transmission:
        flux_star = tools.blackbody_lambda(wl_planet, planet.star_temperature)                  # run_visit.py:201-203
    else:
        raise WFC3SimConfigError("Must give the stellar spectrum if not using transmission spectroscopy")
    stellar_flux_scaled = flux_star * target["flux_scale"]
    wl = wl_planet if transmission else wl_star

    def maybe_file(v):
        return np.loadtxt(path(v)) if isinstance(v, str) else v

    x_ref, y_ref = maybe_file(obs_cfg["x_ref"]), maybe_file(obs_cfg["y_ref"])
    sky_background = maybe_file(obs_cfg["sky_background"])
    exp_start_times = _get(obs_cfg, "exp_start_times", False)
    if exp_start_times:
        exp_start_times = np.loadtxt(path(exp_start_times))
    spatial_scan = obs_cfg["spatial_scan"]
    sample_rate = obs_cfg["sample_rate"] if spatial_scan else False       # ms
    scan_speed = obs_cfg["scan_speed"] if spatial_scan else False         # px/s
    ssv_type = _get(obs_cfg, "ssv_type")
    ssv_gen = None
    if ssv_type:
        ssv_classes = {"sine": scan_speed_varations.SSVSine, "mod-sine": scan_speed_varations.SSVModulatedSine}
        if ssv_type not in ssv_classes:
            raise WFC3SimConfigError("Invalid ssv_type given")                 # run_visit.py:233-245
        ssv_gen = ssv_classes[ssv_type](*obs_cfg["ssv_coeffs"])

    obs = observation.Observation(outdir if os.path.isabs(outdir) else os.path.join(base_dir, outdir),
                                  calibration=cal, device=device, seed=seed)
    obs.setup_detector(det, obs_cfg["NSAMP"], obs_cfg["SAMPSEQ"], obs_cfg["SUBARRAY"])
    obs.setup_grism(chosen_grism)
    obs.setup_target(planet, wl, depth_planet, stellar_flux_scaled, _get(target, "transit_time"),
                     _get(target, "ldcoeffs"), _get(target, "period"), _get(target, "rp"), _get(target, "sma"),
                     _get(target, "inclination"), _get(target, "eccentricity"), _get(target, "periastron"),
                     _get(target, "stellar_radius"))
    obs.setup_visit(obs_cfg["start_JD"] or 0.0, obs_cfg["num_orbits"], exp_start_times)
    obs.setup_reductions(obs_cfg["add_dark"], obs_cfg["add_flat"], obs_cfg["add_gain_variations"],
                         obs_cfg["add_non_linear"], obs_cfg["add_initial_bias"])
    obs.setup_observation(x_ref, y_ref, spatial_scan, scan_speed)
    obs.setup_simulator(sample_rate, obs_cfg["clip_values_det_limits"], _get(general, "threads", 2))
    obs.setup_trends(ssv_gen, obs_cfg["x_shifts"], obs_cfg["x_jitter"], obs_cfg["y_shifts"], obs_cfg["y_jitter"])
    obs.setup_noise_sources(sky_background, obs_cfg["cosmic_rate"], obs_cfg["add_read_noise"],
                            obs_cfg["add_stellar_noise"])
    obs.setup_gaussian_noise(obs_cfg["noise_mean"], obs_cfg["noise_std"])
    coeffs = _get(_get(cfg, "trends", {}), "visit_trend_coeffs")
    if coeffs:
        obs.setup_visit_trend(coeffs)
    return obs


def run(argv=None):
    ap = argparse.ArgumentParser(prog="wayne", description=__doc__.split("\n")[0])
    ap.add_argument("-p", "--parameter_file", required=True)
    ap.add_argument("--calibration", default=None, help="directory holding the WFC3 calibration FITS files")
    ap.add_argument("--device", type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    ap.add_argument("--max-exposures", type=int, default=None, help="only the first M exposures")
    ap.add_argument("--gpus", type=int, default=1, help="start rank processes, one per GPU of this node")
    ap.add_argument("--ranks-per-gpu", type=int, default=1, help="... and this many to a GPU (small sub-arrays)")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)   # ranks meet and report, no GPU work
    ap.add_argument("--resume", action="store_true",
                    help="skip every exposure whose NNNN_raw.fits is already in the output directory, whole and this "
                         "visit's (restart after a failed rank: only the missing files are generated)")
    ap.add_argument("--float64-reads", action="store_true",
                    help="float64 reads from the device (the reference's arithmetic to the file) instead of float32 ones")
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.ranks_per_gpu < 1:
        raise SystemExit("--gpus and --ranks-per-gpu must be at least 1")
    n_ranks = args.gpus * args.ranks_per_gpu
    given = list(sys.argv[1:] if argv is None else argv)
    if n_ranks > 1 and "WORLD_SIZE" not in os.environ and any(a == "--device" or a.startswith("--device=") for a in given):
        # forwarded to every rank it would put them all on one GPU: each rank takes its device from LOCAL_RANK
        raise SystemExit("--device cannot be combined with --gpus / --ranks-per-gpu (a rank's device is its LOCAL_RANK)")
    if n_ranks > 1 and "WORLD_SIZE" not in os.environ:
        # the launcher: nothing in this process has touched a GPU; the children are ranks of a fresh interpreter each
        child = [a for a in (sys.argv[1:] if argv is None else list(argv))]
        cmd = [sys.executable, "-m", "wayne_amd.run_visit"] + child
        extra = {"PYTHONPATH": os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                               [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p])}
        codes, _ = launch.launch_ranks(n_ranks, cmd, extra_env=extra)
        if any(codes):
            raise SystemExit("run_visit: rank exit codes %s" % codes)
        print("run_visit: %d ranks done" % n_ranks)
        return None
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and n_ranks not in (1, world_env):
        raise SystemExit("WORLD_SIZE (%d) != --gpus x --ranks-per-gpu (%d)" % (world_env, n_ranks))
    if args.ranks_per_gpu > 1 and "LOCAL_RANK" in os.environ:
        args.device = int(os.environ["LOCAL_RANK"]) // args.ranks_per_gpu
    if os.environ.get("WAYNE_SHARE_GPU") == "1":       # every rank on device 0: a one-GPU box, or several ranks per GPU on small sub-arrays
        args.device = 0
    if args.dry_run:
        # rehearsal of the N-rank path on a box without N GPUs (tests/test_visit_driver.py): every rank parses the
        # visit, takes its round-robin share and says so; nothing touches a GPU
        with open(args.parameter_file) as f:
            cfg = yaml.safe_load(f)
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        n_exp = int(args.max_exposures or 0) or 16
        from . import visit as _visit
        mine = _visit.shard(n_exp, rank, world)
        print("dry-run rank %d/%d device %d threads %s exposures %s" % (
            rank, world, args.device, os.environ.get("OMP_NUM_THREADS", "-"), ",".join(str(i) for i in mine)), flush=True)
        return None
    launch.pin_to_gpu_numa(args.device)
    with open(args.parameter_file) as f:
        cfg = yaml.safe_load(f)
    base_dir = os.path.dirname(os.path.abspath(args.parameter_file))
    cal = _cal.CalibrationSet.from_directory(args.calibration) if args.calibration else None
    obs = build_observation(cfg, base_dir, cal, args.device)
    if args.max_exposures is not None:
        obs.exp_start_times = obs.exp_start_times[:args.max_exposures]
    if args.float64_reads:
        obs.frame_options["out_dtype"] = np.float64
    os.makedirs(obs.outdir, exist_ok=True)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        # the visit's own two files are written once, by rank 0 (every rank would write the same bytes, but G x R
        # processes truncating and rewriting one path is a race a reader can see)
        shutil.copy2(args.parameter_file, os.path.join(obs.outdir, os.path.basename(args.parameter_file)))
        t, lc = obs.show_lightcurve()
        np.savetxt(os.path.join(obs.outdir, "visit_plan.txt"), np.column_stack([t, lc]), header="JD white_light_model")
    frames = obs.run_observation(rank=rank, world=world, resume=args.resume)
    print("rank %d/%d: wrote %d files to %s%s" % (rank, world, len(frames), obs.outdir,
                                                 " (%d already there)" % len(obs.skipped) if args.resume else ""))
    return obs


if __name__ == "__main__":
    run()
