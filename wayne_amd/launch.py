"""One process per GPU without an external launcher, and CPU affinity of a rank.

`launch_ranks` is what `bench.py --gpus N` and `python -m wayne_amd.run_visit --gpus N` do when they are not
already a rank (no WORLD_SIZE in the environment): the parent -- which has not touched HIP or torch -- starts N
fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set and waits for them.
(A process that has initialised the GPU must never exec another program; starting children from a clean parent
is the safe form.)  Exposures are independent, so the ranks share nothing but the output directory.

`pin_to_gpu_numa` binds the calling process to the CPUs of the NUMA node its GPU hangs off (sysfs, best effort):
the host side of a rank is descriptor building and copies out of pinned memory, both of which want local memory.
"""
import glob
import os
import re
import socket
import subprocess
import sys


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def host_threads_per_rank(world):
    """Threads a rank's host libraries may start: the CPUs this process may run on, shared out over the ranks of the
    node (at least 1).  Without a cap every rank's numpy / OpenMP / BLAS pool sizes itself for the whole machine:
    eight ranks x 256 threads on a 256-CPU host is how the delivered rate of an 8-GPU run falls apart."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    return max(1, cpus // max(1, int(world)))


def rank_env(rank, world, port, extra=None):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    n = str(host_threads_per_rank(world))
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        env.setdefault(var, n)           # (a value the caller exported is respected)
    env.update(extra or {})
    return env


RENDEZVOUS_TIMEOUT_S = 120      # every init_process_group of this package: a missing rank is an error, not a 30 min wait
KILLED_BY_LAUNCHER = -15        # exit code reported for a rank the launcher ended because a sibling had failed


def rendezvous_timeout():
    import datetime
    return datetime.timedelta(seconds=int(os.environ.get("WAYNE_RENDEZVOUS_TIMEOUT_S", RENDEZVOUS_TIMEOUT_S)))


def launch_ranks(n, cmd, extra_env=None, capture_rank0=False, poll_s=0.05, grace_s=5.0):
    """Start `cmd` (a list, e.g. [sys.executable, script, args...]) n times, one per rank, and watch them all:
    the first rank that exits non-zero ends the run -- its siblings (which may be sitting in a rendezvous or a
    barrier that can no longer complete) are terminated, then killed after `grace_s`.
    Returns (exit codes, rank 0's stdout or None); a rank ended by the launcher reports a negative code."""
    import tempfile
    import time
    port = free_port()
    procs = []
    # rank 0's stdout goes to a file, not a pipe: nobody has to drain it while the launcher polls
    cap = tempfile.TemporaryFile(mode="w+") if capture_rank0 else None
    try:
        for r in range(n):
            out = cap if (capture_rank0 and r == 0) else (subprocess.DEVNULL if capture_rank0 else None)
            procs.append(subprocess.Popen(cmd, env=rank_env(r, n, port, extra_env), stdout=out, text=True))
        failed = False
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                failed = True
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
        if failed:
            first = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            sys.stderr.write("launch: rank %d exited with code %d; ending the other ranks\n" % first[0])
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + grace_s
            for p in procs:
                try:
                    p.wait(timeout=max(0.0, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        codes = [p.wait() for p in procs]
        out0 = None
        if cap is not None:
            cap.seek(0)
            out0 = cap.read()
        return codes, out0
    finally:
        for p in procs:          # (an exception in the launcher itself must not leave ranks behind)
            if p.poll() is None:
                p.kill()
        if cap is not None:
            cap.close()


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(device):
    """CPUs local to HIP device `device` from the KFD topology (GPU nodes in node order, the order HIP enumerates
    them in when HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES do not re-map), or None when sysfs does not say."""
    if any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return None                       # the ordinal no longer names a KFD node: do not guess
    gpus = []
    for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(node, "properties")) if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(int(props.get("drm_render_minor", "-1")))
    if device < 0 or device >= len(gpus) or gpus[device] < 0:
        return None
    try:
        text = open("/sys/class/drm/renderD%d/device/local_cpulist" % gpus[device]).read()
    except OSError:
        return None
    cpus = _cpulist(text)
    return cpus or None


def pin_to_gpu_numa(device):
    """sched_setaffinity to the GPU's local CPUs (intersected with what the process may use).  Returns the CPU set
    applied, or None when nothing was done."""
    if os.environ.get("WAYNE_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = gpu_local_cpus(device)
    if not cpus:
        return None
    allowed = os.sched_getaffinity(0) & cpus
    if not allowed or allowed == os.sched_getaffinity(0):
        return None
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:
        return None
    return allowed
