"""Running a visit: exposures sharded round-robin over GPUs, one process per GPU.

Exposures of a visit share no state (the reference only couples them through
its sequential global RNG, observation.py:403-405, which the Philox counters
remove), so exposure i goes to rank i mod G with no collective in the data
path.  Each rank owns one Engine (context + resident
calibration) and writes its own NNNN_raw.fits files (observation.py:427).
"""
import hashlib
import os
import queue
import sys
import threading

import numpy as np

from . import engine as _engine
from .exposure import Exposure, FitsWriterPool
from .exposure_generator import ExposureGenerator


def shard(n_exposures, rank, world):
    """Exposure indices of `rank`: i = rank, rank + world, ... (round-robin keeps
    orbit phase, and so per-exposure cost, balanced across ranks)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_exposures, world))


def descriptor_digest(desc):
    """SHA-1 over everything a descriptor hands to the device (for sharding tests)."""
    h = hashlib.sha1()
    for name, _ in desc._fields_:
        v = getattr(desc, name)
        if isinstance(v, (int, float)):
            h.update(repr((name, v)).encode())
    for a in desc._keep:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


class VisitRunner(object):
    """Generate the exposures `indices` of a synthetic.Visit-like object on one GPU."""

    # exposures in flight: slots 0..DEPTH-1 in rotation, alternating over the context's two streams (an even number
    # keeps the streams balanced; scripts/probe_pipeline.py: 2 -> 826 /s, 3 -> 646, 4 -> 809 for resident descriptors)
    DEPTH = 4

    def __init__(self, visit, device=0, out_dir=None, out_dtype=np.float32, frame_overrides=None, device_lc=False):
        """`device_lc`: hand the device the K + W + 4 numbers of the light-curve model (visit.device_depths)
        instead of a K x W transit-depth matrix computed on the host."""
        self.visit, self.device, self.out_dir = visit, device, out_dir
        self.out_dtype = out_dtype
        self.frame_overrides = frame_overrides or {}
        self.device_lc = device_lc
        self.rng_mode = 2            # WAYNE_RNG_SPLIT
        self._eng = None

    def engine(self):
        if self._eng is None:
            v = self.visit
            self._eng = _engine.get_engine(self.device, v.grism, v.detector, v.calibration, v.NSAMP, v.SAMPSEQ,
                                           v.SUBARRAY,
                                           g102_flat_quirk=bool(self.frame_overrides.get("reference_quirks", False)))
        return self._eng

    def generator(self, i):
        v = self.visit
        return ExposureGenerator(v.detector, v.grism, v.NSAMP, v.SAMPSEQ, v.SUBARRAY, calibration=v.calibration,
                                 device=self.device, seed=v.seed, exposure_index=i,
                                 filename="%04d_raw.fits" % (i + 1))

    def frame_kwargs(self, i):
        over = dict(self.frame_overrides)
        if self.device_lc:
            over["planet_signal"] = self.visit.device_depths(i)
        return self.visit.frame_kwargs(i, **over)

    def descriptor(self, i, eng=None):
        return self.generator(i).build_descriptor(eng, out_dtype=self.out_dtype, rng_mode=self.rng_mode,
                                                  **self.frame_kwargs(i))

    def run(self, indices, keep=False, on_reads=None):
        """Synthesise the given exposures as a pipeline over the context's two HIP streams and pinned
        host buffers: while the kernels of exposure n run on one stream and the device-to-host copy of
        exposure n-1 on the other, the host prepares and uploads exposure n+1 into the next slot.
        `on_reads(i, reads)` is called with a view of the pinned buffer (copy it to keep it);
        keep=True returns {index: copy}; FITS files are written when out_dir is set."""
        eng = self.engine()
        ctx = eng.ctx
        results = {}
        pending = []                       # [(index, slot, generator)] in flight, oldest first
        self._pool = FitsWriterPool() if self.out_dir is not None else None
        try:
            self._run_pipeline(eng, ctx, indices, results, pending, keep, on_reads)
        finally:
            if self._pool is not None:
                self._pool.close()
                self._pool = None
        return results

    def run_resident(self, n, on_reads=None):
        """The same pipeline over descriptors that are ALREADY in slots 0..DEPTH-1 (uploaded by the
        caller): n exposures, kernels + copy to pinned host memory, no host preparation or upload."""
        ctx = self.engine().ctx
        pending = []
        for j in range(n):
            slot = j % self.DEPTH
            if len(pending) == self.DEPTH:
                s_old = pending.pop(0)
                reads = ctx.wait(s_old)
                if on_reads is not None:
                    on_reads(s_old, reads)
            ctx.run(slot)
            ctx.fetch_async(slot)
            pending.append(slot)
        for s_old in pending:
            reads = ctx.wait(s_old)
            if on_reads is not None:
                on_reads(s_old, reads)

    def _run_pipeline(self, eng, ctx, indices, results, pending, keep, on_reads):
        # Two host threads.  A producer prepares descriptors (the K-vectors of an exposure: numpy, the Philox host
        # draws, the light-curve inputs -- no GPU call, no context state); this thread uploads, launches and collects.
        # The C calls on both sides release the interpreter lock (ctypes), so a descriptor's 0.1 ms of host draws and an
        # upload's 0.08 ms of table building overlap the other thread's Python: on the reference's example-visit shape
        # the pipeline is then paced by the device, not by the host.
        ahead = queue.Queue(maxsize=self.DEPTH)
        stop = threading.Event()            # set by this thread when it leaves the loop, for whatever reason

        def put(item):
            """Queue.put that gives up when the consumer has gone (returns False)."""
            while not stop.is_set():
                try:
                    ahead.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for n, i in enumerate(indices):
                    if stop.is_set():
                        return
                    gen = self.generator(i)
                    desc = gen.build_descriptor(eng, out_dtype=self.out_dtype, rng_mode=self.rng_mode,
                                                **self.frame_kwargs(i))
                    if not put((n, i, gen, desc)):
                        return
            except BaseException as e:      # surfaced in the consuming thread
                put(e)
                return
            put(None)

        producer = threading.Thread(target=produce, daemon=True)
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))   # hand the lock over promptly between the two
        producer.start()
        try:
            while True:
                item = ahead.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                n, i, gen, desc = item
                slot = n % self.DEPTH
                if len(pending) == self.DEPTH:          # the slot about to be reused must be drained first
                    self._finish(ctx, pending.pop(0), results, keep, on_reads)
                ctx.upload(slot, desc)
                ctx.run(slot)                  # asynchronous on the slot's stream
                ctx.fetch_async(slot)          # ... followed by its copy to pinned host memory
                pending.append((i, slot, gen))
            while pending:
                self._finish(ctx, pending.pop(0), results, keep, on_reads)
        finally:
            sys.setswitchinterval(old_interval)
            stop.set()              # an error or Ctrl-C on this side: the producer stops after the descriptor it is
            producer.join()         # building, not after the rest of the visit's host work

    def _finish(self, ctx, pending, results, keep, on_reads=None):
        i, slot, gen = pending
        reads = ctx.wait(slot)
        if on_reads is not None:
            on_reads(i, reads)
        if keep:
            results[i] = reads.copy()
        if self.out_dir is not None:
            os.makedirs(self.out_dir, exist_ok=True)
            exp = Exposure(gen.detector, gen.grism, None, gen.exp_info)
            read_dt = np.diff(np.concatenate([[0.0], gen.read_times]))
            own = reads.copy()             # the pinned buffer is reused by the next exposure
            exp.add_read(own[0], {"cumulative_exp_time": 0.0, "read_exp_time": 0.0, "CRPIX1": 0})
            for r in range(len(gen.read_times)):
                exp.add_read(own[r + 1], {"cumulative_exp_time": float(gen.read_times[r]),
                                          "read_exp_time": float(read_dt[r]), "CRPIX1": 0})
            self._pool.submit(exp, self.out_dir, gen.exp_info["filename"])
