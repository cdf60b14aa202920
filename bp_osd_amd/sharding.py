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
    home = local.device
    if dist.get_backend(group) == "nccl" and not local.is_cuda:
        local = local.cuda()  # RCCL moves device tensors only; a numpy-returning decode_fn hands over host memory
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
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).to(home)


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


class StepPipeline:
    """Steps of the sharded decode path with the one exchange step overlapped: decode -> bit-pack -> gather to `dst`.

    This is the loop `bench.py --gpus N` times and the loop tests/test_sharding_cpu.py drives on gloo with a stub
    decoder -- one code path.  The caller supplies the device-side work as callables, all asynchronous except `wait`:

      launch(k, slot)      enqueue the decode of step k on decoder slot `slot` (a slot = one handle + its output buffers)
      pack(slot, buf)      enqueue the bit-packing of that slot's corrections into packed[buf], ordered after the decode
      wait(slot)           block until everything enqueued on that slot has finished
      on_finalised(k, timed)  optional: called once step k's decode has completed (kernel timings are read here)
      on_gathered(k, rows) optional, `dst` only: rows = the gathered packed corrections of step k, rank order, padding
                           rows of short shards removed (the caller must synchronise before reading device tensors)

    `packed` is a pair of tensors [rows_max, words] (rows_max = the largest shard over ranks: short shards are padded,
    a gather needs equal shapes); step k uses packed[k & 1], and a buffer is packed again only after the gather that
    read it has completed (event-guarded on CUDA tensors, synchronous on CPU tensors).  At most `nslots` steps are in
    flight; a slot is reused only after its previous step has been finalised."""

    def __init__(self, nslots, launch, wait, pack=None, packed=None, rows=None, gather=True, group=None, dst=0,
                 on_finalised=None, on_gathered=None):
        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        self.nslots = int(nslots)
        self.launch, self.wait, self.pack = launch, wait, pack
        self.on_finalised, self.on_gathered = on_finalised, on_gathered
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.do_gather = bool(gather) and self.world > 1
        if self.do_gather and self.nslots > 2:
            # packed[k & 1] is gathered when step k is finalised; with three steps in flight step k + 2 would pack into
            # the buffer of step k before that gather has read it
            raise ValueError("with the gather on, at most 2 steps may be in flight (the packed rows are double-buffered)")
        self.packed = packed
        self._gather_marks = []    # per gathered step: (start, end) CUDA events, or a host duration in seconds
        self.pending = []          # (k, timed) enqueued, not yet finalised
        self.gather_done = [None, None]
        self.gather_bufs = None
        self.counts = None
        if self.do_gather:
            if pack is None or packed is None or len(packed) != 2:
                raise ValueError("gathering needs `pack` and a pair of packed buffers")
            rows = packed[0].shape[0] if rows is None else int(rows)
            if rows > packed[0].shape[0]:
                raise ValueError("packed buffers are smaller than the local shard")
            t = torch.tensor([rows], dtype=torch.int64, device=packed[0].device)
            allr = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(allr, t, group=group)
            self.counts = [int(r.item()) for r in allr]
            if max(self.counts) != packed[0].shape[0]:
                raise ValueError(f"packed buffers must have max-shard rows = {max(self.counts)}, not {packed[0].shape[0]}")
            if self.rank == dst:
                self.gather_bufs = [[torch.empty_like(packed[0]) for _ in range(self.world)] for _ in range(2)]

    def _finalise(self, k, timed):
        self.wait(k % self.nslots)
        if self.on_finalised is not None:
            self.on_finalised(k, timed)
        if self.do_gather:
            buf = k & 1
            src = self.packed[buf]
            if src.is_cuda:
                ev0 = self._torch.cuda.Event(enable_timing=True)
                ev0.record()
            else:
                import time as _time
                t0 = _time.perf_counter()
            self._dist.gather(src, self.gather_bufs[buf] if self.rank == self.dst else None, dst=self.dst, group=self.group)
            if src.is_cuda:  # the gather runs on torch's stream: guard the buffer's reuse with an event
                ev = self._torch.cuda.Event(enable_timing=True)
                ev.record()
                self.gather_done[buf] = ev
                if timed:
                    self._gather_marks.append((ev0, ev))
            elif timed:
                self._gather_marks.append(_time.perf_counter() - t0)
            if self.on_gathered is not None and self.rank == self.dst:
                self.on_gathered(k, [b[:c] for b, c in zip(self.gather_bufs[buf], self.counts)])

    def step(self, k, timed=True):
        slot = k % self.nslots
        while len(self.pending) >= self.nslots:  # slot (and its output buffers) of step k - nslots must be free
            self._finalise(*self.pending.pop(0))
        if self.do_gather:
            # the packed buffer of this step must be free BEFORE the decode is enqueued: where the kernels write packed rows
            # themselves (bposd_decode_batch_device_packed) the launch, not the pack step, fills it
            buf = k & 1
            if self.gather_done[buf] is not None:
                self.gather_done[buf].synchronize()
                self.gather_done[buf] = None
        self.launch(k, slot)
        if self.do_gather:
            self.pack(slot, k & 1)
        self.pending.append((k, timed))

    def gather_ms(self):
        """Durations (ms) of the gathers of the timed steps on this rank; call after fence()."""
        out = []
        for g in self._gather_marks:
            out.append(g[0].elapsed_time(g[1]) if isinstance(g, tuple) else 1e3 * g)
        return out

    def drain(self):
        while self.pending:
            self._finalise(*self.pending.pop(0))

    def fence(self):
        """Everything enqueued so far is complete on every rank (the bracket of a timed region)."""
        self.drain()
        if self.world > 1:
            self._dist.barrier(group=self.group)
        if self._torch.cuda.is_available() and self._torch.cuda.is_initialized():
            self._torch.cuda.synchronize()
