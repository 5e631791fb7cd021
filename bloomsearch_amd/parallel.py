"""One-process-per-GPU layer (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The probe shards embarrassingly: block b of a file set lives on rank b % world (round-robin, the
BASELINE sharding), every rank probes its own arena on its own stream, and the only cross-rank step
is a host-side gather of the surviving-block bitsets — no data-path collective.

The single real exchange on this path is the fixed-geometry OR-reduce of partial file-level bitsets
(an extension — the reference rebuilds instead of OR-ing, merge.go:447-453).  RCCL
has no bitwise-OR reduction, so it is an all_gather of the G partial bitsets followed by a local OR
kernel (bsg_or_words_dev on the GPU; torch.bitwise_or on the gloo/CPU test path).
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


def or_allreduce_(words, ctx=None):
    """In-place bitwise-OR all-reduce of a 1-D int64 torch tensor of bitset words (all ranks same length).
    all_gather over RCCL/xGMI (each GPU receives (G-1)/G of the result over its links) + one local OR."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if world == 1:
        return words
    flat = torch.empty(world * words.numel(), dtype=words.dtype, device=words.device)
    dist.all_gather_into_tensor(flat, words)
    gathered = flat.view(world, words.numel())
    if words.is_cuda:
        if ctx is None:
            raise RuntimeError("a bloomgpu Context is required for the device OR (no torch fallback on the GPU path)")
        torch.cuda.current_stream().synchronize()
        ctx.or_words_dev(words.data_ptr(), gathered.data_ptr(), words.numel(), world)
    else:
        acc = gathered[0]
        for r in range(1, world):
            acc = torch.bitwise_or(acc, gathered[r])
        words.copy_(acc)
    return words
