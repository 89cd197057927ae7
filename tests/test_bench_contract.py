"""GPU: bench.py prints exactly one JSON line carrying the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"].startswith("simulated WFC3-IR exposures/sec") and d["unit"] == "exposures/s"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 1.0
    assert d["value"] > 100 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1.0      # one GPU: value = 1000 / ms_per_step
    # the like-for-like and host-pipeline numbers ride in the same line, and never replace `value`
    rep = d["repetitions"]
    assert rep["n"] >= 5 and len(rep["values"]) == rep["n"] and rep["min"] <= d["value"] <= rep["max"]
    for key in ("per_electron", "per_electron_f64", "replay_bit_exact", "out_f64", "two_streams", "delivered", "end_to_end"):
        assert d[key]["unit"] == "exposures/s" and d[key]["value"] > 10, key
    assert d["replay_bit_exact"]["value"] < d["per_electron"]["value"] < d["value"]
    pa = d["psf_apply_replay"]          # the inner drop-in boundary, PCIe-inclusive
    assert 0.01 < pa["ms_per_call"] < 50 and pa["electrons_per_call"] > 1e6 and pa["electrons_per_s"] > 1e9
    # the sustained pass: seconds of back-to-back exposures, reported beside `value`
    sus = d["sustained"]
    assert sus["unit"] == "exposures/s" and sus["seconds"] >= 2.0 and sus["steps"] >= 1000
    assert 0.7 * d["value"] < sus["value"] < 1.2 * d["value"]
    assert d["delivered"]["value"] < d["two_streams"]["value"]
    assert 0 < d["end_to_end"]["frac_of_pcie"] < 1
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] > 1e8
    assert "traffic_source" in d["roofline"]
    # the label says what was timed: the k_ramp instantiation the library reports for the timed slots
    assert r["kernel"] == "k_ramp<float, true, 1, false, true>" and d["dtype"].startswith(r["kernel"])
    assert "f64 ramp" not in d["dtype"] and "all-f32 per-read chain" in d["dtype"]
    # float64 reads are another instantiation with a roofline block of its own (8-byte stores)
    r64 = d["out_f64"]["roofline"]
    assert r64["kernel"].startswith("k_ramp<double, true, 1, false") and r64["bytes_per_launch"] > r["bytes_per_launch"]
    assert abs(r64["frac"] - r64["achieved"] / r64["peak"]) < 1e-9 and 0.1 < r64["frac"] < 1.0
    # the same kernel with its calibration planes evicted before every launch: same bytes, never faster than warm
    cold = r["cold_cache"]
    assert cold["launches_timed"] >= 4 and cold["ms_per_launch"] > 0.8 * r["ms_per_launch"]
    assert abs(cold["frac"] - cold["achieved"] / r["peak"]) < 1e-9 and 0.1 < cold["frac"] < 1.0
    # SURVEY 8(d)'s second leg carries its rank accounting at every N
    for key in ("delivered", "end_to_end"):
        assert d[key]["ranks_reported"] == 1 and list(d[key]["per_rank_exposures_s"]) == ["0"]
    assert len(d["two_streams"]["repetitions"]) == 3
    # one arithmetic per default, and the line says which: the workload names the reads' type and the stream count,
    # two_streams says whose number it is, the library names itself
    import bench
    assert bench.READS_LABEL[False] in d["config"]["workload"] and "one HIP stream" in d["config"]["workload"]
    assert "f32 reads (the default of ExposureGenerator, Observation, VisitRunner and the CLI)" == bench.READS_LABEL[False]
    assert d["two_streams"]["note"].startswith("what VisitRunner (Observation, the CLI) delivers device-side")
    lib_path = os.environ.get("WAYNE_HIP_LIB") or os.path.join(ROOT, "wayne_amd", "libwayne_hip.so")
    assert d["library"] == {"path": os.path.relpath(lib_path, ROOT), "build_flags": "", "abi": 7}
    assert rep["incomplete_exposures"] == 0 and rep["second_runs_in_timed_region"] == 0


def test_every_ramp_instantiation_gets_the_arithmetic_it_runs():
    # `dtype` names the arithmetic of the timed kernels from the instantiation the library reports
    # (wayne_exposure_ramp_variant); over every name select_ramp can emit (wayne_hip.hip): the all-float32 per-read
    # chain exists only for float reads, production math, the table-driven sky in ONE piece and no gaussian-noise stage
    # (k_ramp.h, ramp_body: SKY == 1 && !NOISE)
    sys.path.insert(0, ROOT)
    import bench
    names = []
    for t in ("float", "double"):
        for fast in ("true", "false"):
            for sky in (0, 1, 2):
                for noise in ("false", "true"):
                    pinned = fast == "true" and sky != 0 and noise == "false"
                    if pinned:
                        for allon in (["false", "true"] if (t == "float" and sky == 1) else ["false"]):
                            names.append("k_ramp<%s, %s, %d, %s, %s>" % (t, fast, sky, noise, allon))
                    else:
                        names.append("k_ramp_wide<%s, %s, %d, %s>" % (t, fast, sky, noise))
    assert len(names) == 25
    f32_chain = [n for n in names if "all-f32 per-read chain" in bench.dtype_label(n, n.split("<")[1].startswith("double"))]
    assert sorted(f32_chain) == ["k_ramp<float, true, 1, false, false>", "k_ramp<float, true, 1, false, true>"]
    for n in names:
        label = bench.dtype_label(n, "<double" in n)
        assert label.startswith(n) and label.endswith("f64 reads" if "<double" in n else "f32 reads")
        if n not in f32_chain:
            assert "f64 cumulative sum" in label, n
    # the source says the same: the production chain is taken when SKY == 1 && !NOISE, for float reads, in fast math
    src = open(os.path.join(ROOT, "wayne_amd", "csrc", "k_ramp.h")).read()
    assert "SKY == 1 && !NOISE" in src


def test_launcher_refuses_a_rank_count_mismatch():
    # CPU: under a launcher that set WORLD_SIZE, --gpus must agree with it (never a silent n_gpus = 1)
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE (2) != --gpus (4)" in (out.stderr + out.stdout)


def test_launcher_rendezvous_at_eight_ranks():
    # CPU: `bench.py --gpus 8 --dry-run` -- the self-launching path at the rank count of the driver's scaling run: the
    # parent starts 8 fresh rank processes, they meet over gloo (barrier, max-reduce, gather) and rank 0's line is
    # relayed; no GPU work is rehearsed by this (a one-GPU box admits at most 6 processes on its card)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip())
    assert d == {"dry_run": True, "n_gpus": 8, "ranks_reported": 8, "max_rank": 7.0, "value": None}


def test_the_drivers_torchrun_command_leaves_one_line_on_stdout():
    # CPU: the driver launches N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    # 127.0.0.1 --master-port P bench.py --gpus N ...` and reads ONE JSON line from the job's stdout -- on which every rank's
    # descriptor 1 ends up, and gloo announces "[Gloo] Rank 0 is connected to 1 peer ranks ..." there from C++ (seen on
    # the GPU box in round 5: three lines instead of one).  bench.py keeps descriptor 1 aside for its line and sends
    # whatever else is written to it to stderr
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    assert json.loads(lines[0]) == {"dry_run": True, "n_gpus": 2, "ranks_reported": 2, "max_rank": 1.0, "value": None}


def test_launcher_starts_n_ranks_and_fails_loudly_without_gpus():
    # CPU: `--gpus 2` without WORLD_SIZE starts two rank processes; without a GPU both refuse, and the parent
    # reports the failure instead of printing a one-GPU line
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert "rank exit codes" in out.stderr and "{" not in out.stdout


def test_launcher_fails_fast_when_one_rank_dies_before_the_rendezvous():
    # CPU: rank 1 of 4 exits with code 3 before init_process_group.  The launcher must end the other three -- which
    # sit in a rendezvous that can no longer complete -- and report the failure at once, not after gloo's timeout
    # (the first real SCALE run would otherwise be recorded as a kill at the driver's time limit)
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["WAYNE_DRY_RUN_FAIL_RANK"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    dt = time.time() - t0
    assert out.returncode != 0 and dt < 30.0, (out.returncode, dt)
    assert "rank 1 exited with code 3" in out.stderr and "rank exit codes" in out.stderr
    assert "{" not in out.stdout                        # no line that could be mistaken for a measurement


def test_launch_ranks_reports_codes_and_kills_siblings(tmp_path):
    # the launcher itself: a rank that fails late (after the others have finished) and one that fails while the others
    # would run forever
    import time
    from wayne_amd import launch
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK']); mode = sys.argv[1]\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 0: print('hello from rank 0', flush=True)\n"
                      "if mode == 'ok': sys.exit(0)\n"
                      "if mode == 'late': time.sleep(0.5 if r == 2 else 0.0); sys.exit(7 if r == 2 else 0)\n"
                      "if r == 1: time.sleep(0.3); sys.exit(3)\n"
                      "time.sleep(600)\n")
    codes, out0 = launch.launch_ranks(3, [sys.executable, str(script), "ok"], capture_rank0=True)
    assert codes == [0, 0, 0] and out0 == "hello from rank 0\n"
    codes, _ = launch.launch_ranks(3, [sys.executable, str(script), "late"], capture_rank0=True)
    assert codes == [0, 0, 7]
    t0 = time.time()
    codes, out0 = launch.launch_ranks(3, [sys.executable, str(script), "hang"], capture_rank0=True)
    assert time.time() - t0 < 15.0
    assert codes[1] == 3 and codes[0] < 0 and codes[2] < 0 and out0 == "hello from rank 0\n"
