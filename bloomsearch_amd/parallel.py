"""One-process-per-GPU layer (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The probe shards embarrassingly: block b of a file set lives on rank b % world (round-robin, the
BASELINE sharding), every rank probes its own arena on its own stream, and the only cross-rank step
is a host-side gather of the surviving-block bitsets — no data-path collective.

The single real exchange on this path is the fixed-geometry OR-reduce of partial file-level bitsets
(an extension — the reference rebuilds instead of OR-ing, merge.go:447-453).  RCCL
has no bitwise-OR reduction, so it is reduce-scatter + all-gather with the OR done locally: slice j of
every partial travels to rank j, is OR-ed there (bsg_or_words_dev on the GPU; torch.bitwise_or on the
gloo/CPU test path), and the reduced slices are all-gathered: 2 (G - 1) / G x S on the wire per GPU.
"""
from __future__ import annotations

import numpy as np


def shard_block_ids(n_blocks: int, rank: int, world: int) -> np.ndarray:
    """Global ids of the blocks rank holds: rank, rank + world, ..."""
    return np.arange(rank, n_blocks, world, dtype=np.int64)


def interleave_survivors(parts, n_blocks: int) -> np.ndarray:
    """parts[r]: u64 [Q, ceil(n_local_r / 64)] survivors of rank r's shard (local block lb == global lb*world + r).
    -> u64 [Q, ceil(n_blocks / 64)] in global block order."""
    world = len(parts)
    nq = parts[0].shape[0]
    bits = np.zeros((nq, (n_blocks + 63) // 64 * 64), dtype=np.uint8)
    for r, p in enumerate(parts):
        n_local = len(shard_block_ids(n_blocks, r, world))
        if n_local == 0:
            continue
        local = np.unpackbits(np.ascontiguousarray(p, dtype="<u8").view(np.uint8).reshape(nq, -1), axis=1, bitorder="little")
        bits[:, r: r + n_local * world: world] = local[:, :n_local]
    return np.packbits(bits, axis=1, bitorder="little").view("<u8").reshape(nq, -1).astype(np.uint64)


def gather_survivors(local: np.ndarray, n_blocks: int, dst: int = 0):
    """Host-side gather of per-rank survivor bitsets to rank dst (returns None elsewhere)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    g_max = ((n_blocks + world - 1) // world + 63) // 64
    buf = np.zeros((local.shape[0], g_max), dtype=np.int64)
    buf[:, : local.shape[1]] = local.view(np.int64)
    t = torch.from_numpy(buf).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, outs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r, o in enumerate(outs):
        n_local = len(shard_block_ids(n_blocks, r, world))
        parts.append(o.cpu().numpy().view(np.uint64)[:, : max((n_local + 63) // 64, 1)])
    return interleave_survivors(parts, n_blocks)


def or_slice_len(n_words: int, world: int) -> int:
    """Words per slice of the OR all-reduce's slice schedule: ceil(n_words / world)."""
    return (n_words + world - 1) // world


def or_allreduce_wire_bytes(n_words: int, world: int) -> int:
    """Bytes one rank receives (and sends) in the slice schedule: (world - 1) slices in the exchange + (world - 1) reduced
    slices in the all-gather = 2 (world - 1) / world x S, against (world - 1) x S for an all-gather of the full partials."""
    return 0 if world <= 1 else 2 * (world - 1) * or_slice_len(n_words, world) * 8


def or_allreduce_(words, ctx=None):
    """In-place bitwise-OR all-reduce of a 1-D int64 torch tensor of bitset words (all ranks same length).
    RCCL has no bitwise-OR reduction, so it is reduce-scatter + all-gather with the OR spelled out (the schedule
    bsg_or_allreduce runs inside the library, csrc/comm_api.inc): all_to_all of the `world` slices of every partial
    (rank j receives slice j of everyone), a local OR of the received slices (bsg_or_words_dev on the GPU; torch.bitwise_or
    on the gloo/CPU test path), all_gather of the reduced slices."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if world == 1:
        return words
    rank = dist.get_rank()
    n = words.numel()
    L = or_slice_len(n, world)
    work = torch.zeros(world * L, dtype=words.dtype, device=words.device)
    work[:n] = words
    inbox = torch.empty_like(work)
    dist.all_to_all_single(inbox, work)                      # inbox[j L : (j + 1) L] = slice `rank` of rank j's partial
    mine = work[rank * L: (rank + 1) * L]
    if words.is_cuda:
        if ctx is None:
            raise RuntimeError("a bloomgpu Context is required for the device OR (no torch fallback on the GPU path)")
        torch.cuda.current_stream().synchronize()
        # the inbox holds my own slice too (slot `rank`): OR-ing it in again is harmless
        ctx.or_words_dev(mine.data_ptr(), inbox.data_ptr(), L, world)
    else:
        acc = inbox.view(world, L)[0]
        for r in range(1, world):
            acc = torch.bitwise_or(acc, inbox.view(world, L)[r])
        mine.copy_(acc)
    out = torch.empty_like(work)
    dist.all_gather_into_tensor(out, mine.contiguous())
    words.copy_(out[:n])
    return words
