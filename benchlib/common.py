"""bench.py's shared pieces: the synthetic generator's worker pool, the shared host segment of the host-side gather, the dispatch-timestamp
bookkeeping and the Prober that drives bsg_probe_many over the timed steps."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy


def _gen_block(args):
    from bloomsearch_amd import synth
    b, rows, seed = args
    return synth.block_entry_sets(b * rows, rows, seed)


_POOL = None      # worker processes of the synthetic generator, forked ONCE at the start of main(): before torch, the HIP runtime, RCCL and
_POOL_N = 0       # their threads exist in this process (a fork taken later would copy a process that holds device and collective state)


def start_pool(workers):
    global _POOL, _POOL_N
    if workers > 1 and _POOL is None:
        import multiprocessing as mp
        _POOL, _POOL_N = mp.get_context("fork").Pool(workers), workers


def stop_pool():
    global _POOL
    if _POOL is not None:
        _POOL.close()
        _POOL.join()
        _POOL = None


def pool_map(fn, jobs, workers):
    if _POOL is None or workers <= 1 or len(jobs) < 8:
        return [fn(j) for j in jobs]
    return _POOL.map(fn, jobs, chunksize=max(1, len(jobs) // (_POOL_N * 4)))


def generate_blocks(block_ids, rows, seed, workers):
    return pool_map(_gen_block, [(int(b), rows, seed) for b in block_ids], workers)


def _gen_rows(args):
    from bloomsearch_amd import synth
    b, rows, seed = args
    rs = synth.rows_json(b * rows, rows, seed)
    return b"".join(rs), np.asarray([len(r) for r in rs], dtype=np.uint32)


def measured_copy_gbps(log):
    """the roofline fraction is quoted against the vendor's 8 TB/s AND against what a plain device-to-device copy
    reaches on this very box (1 GiB hipMemcpyAsync D2D through torch, bytes read + bytes written over the copy's own events)."""
    import torch
    try:
        n = 1 << 30
        a = torch.zeros(n, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        e1.synchronize()
        gbps = 2.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        log("device-to-device copy of 1 GiB: %.0f GB/s (read + write)" % gbps)
        return gbps
    except Exception as exc:  # noqa: BLE001 - a calibration figure, never fatal
        log("copy bandwidth not measured: %r" % (exc,))
        return None


def COLL_DEVICE():
    """Where the harness' own small collectives live: the GPU under RCCL, the host under the gloo lab mode."""
    import torch.distributed as dist
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


class Ring:
    """Hands out successive slices of a buffer, wrapping around: the output slots of successive calls (only the last ones are kept)."""

    def __init__(self, buf):
        self.buf, self.o, self.last = buf, 0, 0

    def __getitem__(self, sl):
        n = sl.stop - sl.start
        if self.o + n > len(self.buf):
            self.o = 0
        self.last = self.o
        self.o += n
        return self.buf[self.last: self.last + n]


class SharedHost:
    """One POSIX shared-memory segment mapped by every rank; rank r's survivors are DMA-ed into slice r
    (page-locked with bsg_host_register), so rank 0 reads every shard's bitsets from host memory after the
    closing barrier: the host-side gather, without a collective and with every GPU on its own PCIe link."""

    def __init__(self, ctx, bytes_per_rank, rank, world, tag):
        import mmap
        self.ctx, self.rank, self.world = ctx, rank, world
        self.bytes_per_rank = (int(bytes_per_rank) + 4095) // 4096 * 4096
        self.path = "/dev/shm/bsg_%s_%s" % (os.environ.get("MASTER_PORT", "0"), tag)
        total = self.bytes_per_rank * world
        if rank == 0:
            with open(self.path, "wb") as f:
                f.truncate(total)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        self.f = open(self.path, "r+b")
        self.mm = mmap.mmap(self.f.fileno(), total)
        self.all = np.frombuffer(self.mm, dtype=np.uint64)
        w = self.bytes_per_rank // 8
        self.mine = self.all[rank * w: (rank + 1) * w]
        self.mine[:] = 0                      # touch the pages before locking them
        try:
            ctx.host_register(self.mine)
            self.registered = True
        except Exception as exc:              # noqa: BLE001 - the copies still work (staged by the driver), only slower
            print("[bench] hipHostRegister of the shared segment failed: %r" % (exc,), file=sys.stderr, flush=True)
            self.registered = False

    def part(self, r):
        w = self.bytes_per_rank // 8
        return self.all[r * w: (r + 1) * w]

    def close(self):
        if self.registered:
            self.ctx.host_unregister(self.mine)
        del self.mine, self.all
        try:
            self.mm.close()
        except BufferError:
            pass
        self.f.close()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


C5_TOTAL_FILTERS = 10000     # BASELINE configs[4]: "OR-reduce of 10 000 block bloom filters", shared over the ranks at N > 1
CLOCK_NOTE = ("K steps between barrier + device synchronize on both sides, MAX over ranks; the barrier is HostBarrier (epoch flags in "
              "a shared-memory segment, ~1 us) entered after each rank's own torch.cuda.synchronize(); rank_clock = the same region up to "
              "the rank's own synchronize (closing barrier outside)")


def effective_cpus():
    """(CPUs this process may use, logical CPUs of the box): the affinity mask cut by the cgroup's CPU quota (cpu.max of cgroup v2,
    cpu.cfs_quota_us / cpu.cfs_period_us of v1).  A 256-thread pool on a 16-CPU quota measures the scheduler, not the loop."""
    logical = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = logical
    quota = None
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            quota = float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n), logical


class HostBarrier:
    """The barrier that closes (and tightens the opening of) a timed region at N > 1: one 64-byte slot per rank in a POSIX
    shared-memory segment, each rank publishes its epoch and spins until every slot has reached it.  All ranks of the job are
    processes of ONE node (the contract), so no collective is needed to learn that everybody's device has drained: a `nccl`
    dist.barrier() is an all-reduce kernel plus a stream wait — tens of microseconds inside a region of ~200 us at the 8-rank
    shard — and this one costs about a microsecond.  world == 1: a no-op."""

    def __init__(self, rank, world):
        self.rank, self.world, self.epoch = rank, world, 0
        if world <= 1:
            return
        import mmap
        import torch.distributed as dist
        self.path = "/dev/shm/bsg_%s_barrier" % os.environ.get("MASTER_PORT", "0")
        if rank == 0:
            with open(self.path, "wb") as f:
                f.truncate(64 * world)
        dist.barrier()
        self.f = open(self.path, "r+b")
        self.mm = mmap.mmap(self.f.fileno(), 64 * world)
        self.slots = np.frombuffer(self.mm, dtype=np.int64)[:: 8]          # slot r = word 8 r: one cache line per rank

    def wait(self, timeout_s=60.0):
        if self.world <= 1:
            return
        self.epoch += 1
        self.slots[self.rank] = self.epoch
        t_end, e, s = time.perf_counter() + timeout_s, self.epoch, self.slots
        while int(s.min()) < e:
            if time.perf_counter() > t_end:
                raise RuntimeError("HostBarrier: rank %d waited %.0fs at epoch %d (slots %s)" % (self.rank, timeout_s, e, s.tolist()))

    def close(self):
        if self.world <= 1:
            return
        import torch.distributed as dist
        del self.slots
        try:
            self.mm.close()
        except BufferError:
            pass
        self.f.close()
        dist.barrier()
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


def kernel_stats(tm, n_terms, probe_kernel="k_probe_terms"):
    """Per-kernel roofline inputs from the library's dispatch timestamps (bsg_timing).  probe_kernel: "k_probe_terms_many" for batches with
    more than 128 distinct terms of one kind."""
    out = {}
    if tm.n_probes:
        ms = tm.ms_terms_kernel / tm.n_probes
        by = tm.stream_bytes / tm.n_probes + 33 * n_terms
        out[probe_kernel] = {"samples": int(tm.n_probes), "arenas_per_launch": tm.n_probe_arenas / tm.n_probes,
                                "kernel_ms": ms, "algorithmic_bytes_per_launch": by, "achieved": by / ms / 1e6,
                                "frac": by / ms / 1e6 / HBM_PEAK_GBPS}
    if tm.n_fused:
        ms = tm.ms_fused_kernel / tm.n_fused
        by = tm.fused_stream_bytes / tm.n_fused + 33 * n_terms
        out["k_probe_fused"] = {"samples": int(tm.n_fused), "arenas_per_launch": tm.n_fused_arenas / tm.n_fused,
                                "kernel_ms": ms, "algorithmic_bytes_per_launch": by, "achieved": by / ms / 1e6,
                                "frac": by / ms / 1e6 / HBM_PEAK_GBPS,
                                "note": "probe role of group i + program evaluation of group i-1 in one dispatch; bytes count the probe role's bitsets only"}
    if tm.n_folded:
        ms = tm.ms_folded_kernel / tm.n_folded
        by = tm.folded_stream_bytes / tm.n_folded + 33 * n_terms
        out["k_probe_eval"] = {"samples": int(tm.n_folded), "arenas_per_launch": tm.n_folded_arenas / tm.n_folded,
                               "kernel_ms": ms, "algorithmic_bytes_per_launch": by, "achieved": by / ms / 1e6,
                               "frac": by / ms / 1e6 / HBM_PEAK_GBPS,
                               "note": "k_probe_terms with the program evaluation folded in per tile of blocks: ONE dispatch per group of arenas streams "
                                       "the bitsets AND writes the survivors; bytes count the bitsets + term table only (the survivor words it also "
                                       "writes are not credited)"}
    if tm.n_eval:
        out["k_eval_programs"] = {"samples": int(tm.n_eval), "kernel_ms": tm.ms_eval_kernel / tm.n_eval}
    return out


def dominant_kernel(tm, probe_kernel="k_probe_terms"):
    """The kernel that streamed the bitsets of a timed region: the folded dispatch when the batch takes it."""
    best = max((tm.ms_folded_kernel, "k_probe_eval"), (tm.ms_fused_kernel, "k_probe_fused"), (tm.ms_terms_kernel, probe_kernel))
    return best[1]


def merge_timing(a, b):
    """Sum of two bsg_timing records (same launch shape)."""
    from bloomsearch_amd._lib import Timing
    t = Timing()
    for f, _ in Timing._fields_:
        setattr(t, f, getattr(a, f) + getattr(b, f))
    return t


# flags added to every bsg_probe_many_rows call of the bench (BSG_BENCH_ROWS_DENSE=1: the dense slot layout of rounds 4-5)
ROWS_PACKED = os.environ.get("BSG_BENCH_ROWS_DENSE", "0") != "1"
ROWS_FLAGS = 8 if ROWS_PACKED else 0          # _lib.PROBE_ROWS_PACKED


class Prober:
    """Drives bsg_probe_many over a list of steps (each step = the arena ids probed once)."""

    def __init__(self, ctx, bid, world, log, barrier=None):
        self.ctx, self.bid, self.world, self.log = ctx, bid, world, log
        self.barrier = barrier if barrier is not None else HostBarrier(0, 1)

    def sync_all(self):
        """Opens a timed region: collective barrier + device synchronize (the contract), then the host barrier so that every rank's
        clock starts within a microsecond of the others' (ranks leave a nccl barrier tens of microseconds apart)."""
        import torch
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        self.barrier.wait()

    def plan(self, steps, per_call, out=None, words_per_step=0, hdr=None, hdr_per_step=0):
        """The calls of a run, arguments marshalled (arena id arrays, output slices): `per_call` consecutive steps share one
        bsg_probe_many call.  Built before the clock starts: listing arena ids is the caller's bookkeeping, not the step.
        hdr given: the calls are bsg_probe_many_rows (survivor rows: header + ids / words where needed, written by the device)."""
        calls, o, ho = [], 0, 0
        for i in range(0, len(steps), per_call):
            ids = np.ascontiguousarray([a for st in steps[i: i + per_call] for a in st], dtype=np.uint64)
            dst = hd = None
            if out is not None:
                n = words_per_step * len(steps[i: i + per_call])
                dst = out[o: o + n]
                o += n
            if hdr is not None:
                n = hdr_per_step * len(steps[i: i + per_call])
                hd = hdr[ho: ho + n]
                ho += n
            # (pointers taken here: what is left for the timed region is the C call itself)
            calls.append((ids, dst, hd, ids.ctypes.data, len(ids), None if dst is None else dst.ctypes.data, None if hd is None else hd.ctypes.data))
        return calls

    def run(self, calls, flags):
        """Enqueue every call of a plan: one bsg_probe_many / bsg_probe_many_rows C call each, arguments marshalled by plan()."""
        from bloomsearch_amd import _lib
        L, h, bid, fl = self.ctx.L, self.ctx.h, self.bid, flags | _lib.PROBE_ASYNC
        for _ids, _dst, _hd, p_ids, n, p_dst, p_hd in calls:
            # (rows in their packed form: a run's LIST / DENSE payloads leave as one contiguous stretch instead of one PCIe write per row)
            rc = L.bsg_probe_many(h, p_ids, n, bid, fl, p_dst) if p_hd is None else L.bsg_probe_many_rows(h, p_ids, n, bid, fl | ROWS_FLAGS, p_dst, p_hd)
            if rc:
                self.ctx._check(rc)

    def measure(self, make_step, steps, warmup, per_call, timed=True, out=None, words_per_step=0, nofuse=False, hdr=None, hdr_per_step=0):
        """The contract's region: barrier + device synchronize on both sides of exactly `steps` steps, MAX over ranks.  Returned time
        = the region WITH the closing barrier inside (every rank's K steps are complete in HBM and every rank knows it); the closing
        barrier is HostBarrier, entered after the rank's own torch.cuda.synchronize().  self.last_closing keeps the rank's own clock
        (closing barrier outside) beside it.
        The rotation index runs on from the warm-up into the timed steps (step i of the region is make_step(warmup + i)): the timed
        region never starts on the arenas the warm-up just left in the 256 MiB Infinity Cache."""
        import torch
        from bloomsearch_amd import _lib
        flags = (_lib.PROBE_TIMED if timed else 0) | (_lib.PROBE_NOFUSE if nofuse else 0)
        self.ctx.set_timed_stride(1)
        self.run(self.plan([make_step(i) for i in range(warmup)], per_call, out, words_per_step, hdr, hdr_per_step), flags)
        self.ctx.sync()
        self.ctx.timing_read(reset=True)
        calls = self.plan([make_step(warmup + i) for i in range(steps)], per_call, out, words_per_step, hdr, hdr_per_step)
        self.sync_all()
        t0 = time.perf_counter()
        self.run(calls, flags)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()           # this rank's K steps are done: the device-wide wait covers the library's streams
        dt_rank = time.perf_counter() - t0
        self.barrier.wait()                # ... and so are everybody else's
        dt = time.perf_counter() - t0
        self.ctx.sync()                    # (the library's own bookkeeping of finished copies, outside the clock)
        dt_local = dt_rank
        if self.world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt, dt_rank], dtype=torch.float64, device=COLL_DEVICE())
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, dt_rank = float(t[0].item()), float(t[1].item())
        self.last_closing = {"ms_per_step_rank_clock": dt_rank / steps * 1e3, "ms_per_step_this_rank": dt_local / steps * 1e3,
                             "closing_barrier_us": (dt - dt_rank) * 1e6}
        self.log("%d steps: host enqueue %.2f us/step, wall %.2f us/step (max over ranks, closing barrier inside; %.2f on the ranks' own clocks)%s"
                 % (steps, t_enq / steps * 1e6, dt / steps * 1e6, dt_rank / steps * 1e6,
                    " (survivors delivered to host memory)" if out is not None else ""))
        return dt, self.ctx.timing_read()


