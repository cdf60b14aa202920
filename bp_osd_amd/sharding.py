"""Multi-GPU data-parallel plumbing for the decode path.

Syndromes are independent, so a batch is split into contiguous shards, one per rank (one process per
GPU, decoder state replicated at construction).  The only exchange step is the final gather of the
corrections to rank 0 -- `torch.distributed.gather`, which is RCCL over xGMI with backend "nccl" on
ROCm and gloo in the CPU tests.  No collective is needed anywhere else; LER counters are three
integers reduced with one tiny all-reduce.
"""
from __future__ import annotations


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) slice of `total` items for `rank` (first `total % world` ranks get
    one extra item)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_to_root(local, dst: int = 0, group=None):
    """Gather equally- or unequally-sized row shards (dim 0) to `dst`; returns the concatenation on
    `dst` (rank order = shard order) and None elsewhere."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows, group=group)
    counts = [int(r.item()) for r in all_rows]
    mx = max(counts)
    if local.shape[0] < mx:  # pad to a common shape: gather needs equal sizes
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def reduce_counts(counts, group=None):
    """Sum a small list of integer counters (e.g. shots, logical failures, BP convergences) over ranks."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [int(c) for c in counts]
    t = torch.tensor([int(c) for c in counts], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return [int(x) for x in t.cpu().tolist()]


def decode_sharded(decode_fn, syndromes, group=None, dst: int = 0):
    """Decode a global batch held identically on every rank: each rank decodes its contiguous shard
    with `decode_fn(shard) -> uint8 tensor/ndarray [rows, n]`, results are gathered to `dst`."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(syndromes), rank, world)
    out = decode_fn(syndromes[lo:hi])
    if isinstance(out, np.ndarray):
        out = torch.from_numpy(np.ascontiguousarray(out))
    return gather_to_root(out, dst=dst, group=group)
