"""world_size-2 gloo tests of the N>1 plumbing (runs on CPU): round-robin block sharding, the host-side
gather/interleave of survivor bitsets, and the OR all-reduce of partial bitsets (slice exchange + local OR + all-gather).  The per-rank
"device" results are stood in by the oracle here (the GPU kernels cannot run on this host); what is under
test is the distributed layer in bloomsearch_amd/parallel.py."""
import os
import socket

import numpy as np
import pytest

from bloomsearch_amd import parallel as P


def test_interleave_survivors_inverse_of_sharding():
    rng = np.random.default_rng(0)
    for n_blocks, world in ((1, 2), (63, 2), (64, 2), (130, 2), (1000, 8), (5, 8)):
        nq = 7
        bits = rng.integers(0, 2, size=(nq, n_blocks), dtype=np.uint8)
        parts = []
        for r in range(world):
            ids = P.shard_block_ids(n_blocks, r, world)
            lb = np.zeros((nq, max((len(ids) + 63) // 64, 1) * 64), dtype=np.uint8)
            lb[:, : len(ids)] = bits[:, ids]
            parts.append(np.packbits(lb, axis=1, bitorder="little").view("<u8").astype(np.uint64))
        full = np.zeros((nq, (n_blocks + 63) // 64 * 64), dtype=np.uint8)
        full[:, :n_blocks] = bits
        want = np.packbits(full, axis=1, bitorder="little").view("<u8").astype(np.uint64)
        assert np.array_equal(P.interleave_survivors(parts, n_blocks), want)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_blocks, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bloomsearch_amd import query as Q
        from bloomsearch_amd.arena import entry_sets_from_strings, plan_blocks
        from oracle import oracle as O
        from tests.helpers import oracle_terms, oracle_words

        def block(b):
            toks = ["t%d" % (b % 7), "u%d" % (b % 5), "all"] + (["rare"] if b % 41 == 3 else [])
            return entry_sets_from_strings(["f%d" % (b % 3)], toks, ["g::" + x for x in toks])
        exprs = [Q.Token("rare"), Q.And(Q.Token("t3"), Q.Token("u2")), Q.Or(Q.Field("f1"), Q.FieldToken("g", "u4")), None, Q.Token("nope")]
        cb = Q.compile_queries(exprs)
        ops, poff, _ = cb.arrays()
        terms = oracle_terms(cb).view(O.TERM_DTYPE)
        # this rank's shard (stand-in for its GPU arena + probe)
        ids = P.shard_block_ids(n_blocks, rank, world)
        plan = plan_blocks([block(int(b)) for b in ids], 0.01)
        local = O.probe_batch(oracle_words(plan), plan.desc.view(O.DESC_DTYPE), terms, ops, poff)
        got = P.gather_survivors(local, n_blocks, dst=0)
        ok = True
        if rank == 0:
            gplan = plan_blocks([block(b) for b in range(n_blocks)], 0.01)
            want = O.probe_batch(oracle_words(gplan), gplan.desc.view(O.DESC_DTYPE), terms, ops, poff)
            ok = np.array_equal(got, want)
        # fixed-geometry OR-reduce: partial file-level bitsets built at one (m, k) from each rank's entries
        universe = ["tok%d" % i for i in range(5000)]
        m, k = O.estimate_parameters(len(universe), 0.001)
        part = O.Filter(m, k)
        for i in range(rank, len(universe), world):
            part.add(universe[i])
        t = torch.from_numpy(part.words.view(np.int64).copy())
        P.or_allreduce_(t)
        full = O.Filter(m, k)
        for s in universe:
            full.add(s)
        ok = ok and np.array_equal(t.numpy().view(np.uint64), full.words)
        # the slice schedule with word counts that do not divide by the world size (a padded last slice), incl. 1 word
        for nw in (1, 2, 3, 1001):
            rng = np.random.default_rng(1000 + nw)
            parts = [rng.integers(0, 1 << 62, size=nw, dtype=np.int64) & rng.integers(0, 1 << 62, size=nw, dtype=np.int64) for _ in range(world)]
            t = torch.from_numpy(parts[rank].copy())
            P.or_allreduce_(t)
            want = parts[0]
            for p2 in parts[1:]:
                want = want | p2
            ok = ok and np.array_equal(t.numpy(), want)
        ok = ok and P.or_allreduce_wire_bytes(1001, 2) == 2 * 1 * 501 * 8 and P.or_allreduce_wire_bytes(8000, 8) == 2 * 7 * 1000 * 8
        q.put((rank, bool(ok)))
    except Exception as exc:   # report instead of letting the parent wait out its timeout
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_blocks", [129, 1000])
def test_two_rank_gather_and_or_reduce_gloo(n_blocks):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_blocks, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, True), (1, True)]


def _barrier_worker(rank, world, port, q):
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from benchlib.common import HostBarrier
        hb = HostBarrier(rank, world)
        ok = True
        for i in range(50):
            # the late rank alternates: nobody may leave round i before the late rank has entered it
            late = i % world
            if rank == late:
                time.sleep(0.002)
            t_in = time.monotonic()
            hb.wait()
            t_out = time.monotonic()
            ins = [None] * world
            dist.all_gather_object(ins, (t_in, t_out))
            ok = ok and min(o for _, o in ins) >= max(i_ for i_, _ in ins)       # CLOCK_MONOTONIC is one clock for all processes of a node
        hb.close()
        q.put((rank, bool(ok) and (rank != 0 or not os.path.exists(hb.path))))       # rank 0 removes the segment after everybody closed it
    except Exception as exc:
        q.put((rank, repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


def test_host_barrier_two_ranks_gloo():
    """bench.py's closing barrier at N > 1 (benchlib.common.HostBarrier: epoch flags in a shared-memory segment)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_barrier_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, True), (1, True)]
