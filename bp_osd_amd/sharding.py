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

      launch(k, slot, buf) enqueue the decode of step k on decoder slot `slot` (a slot = one handle + its output buffers);
                           `buf` = the packed buffer of the step (None without the gather) -- where the kernels write
                           bit-packed rows themselves the launch fills it
      pack(slot, buf)      enqueue the bit-packing of that slot's corrections into packed[buf], ordered after the decode
      wait(slot)           block until everything enqueued on that slot has finished
      on_finalised(k, timed)  optional: called once step k's decode has completed (kernel timings are read here)
      on_gathered(k, rows) optional, `dst` only: rows = the gathered packed corrections of step k, rank order, padding
                           rows of short shards removed; called when the gather is known to be complete (the caller
                           must synchronise before reading device tensors)
      gather_fn(src, bufs_or_None) optional: issues the gather and returns a handle with `.synchronize()` (or `.wait()`);
                           default: `dist.gather` -- on CUDA tensors guarded by an event on torch's stream, on CPU
                           tensors `async_op=True`

    `packed` is a list of `nbuf` tensors [rows_max, words] (rows_max = the largest shard over ranks: short shards are
    padded, a gather needs equal shapes); step k uses packed[k % nbuf], and a buffer is written again only after the
    gather that read it has completed.  The gather of step k is issued when step k is finalised, i.e. just before step
    k + nslots is launched: with nbuf = nslots the launch of step k + nslots would wait for the gather issued a moment
    before (one collective round trip on the critical path of every step -- rounds 1-4's double buffer); nbuf = nslots + 2
    gives every gather two whole steps before its buffer is needed again, so a launch never waits for one.  nbuf < nslots
    is refused (step k + nbuf would overwrite rows whose gather has not even been issued).  At most `nslots` steps are
    in flight; a slot is reused only after its previous step has been finalised."""

    def __init__(self, nslots, launch, wait, pack=None, packed=None, rows=None, gather=True, group=None, dst=0,
                 on_finalised=None, on_gathered=None, gather_fn=None):
        import inspect

        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        self.nslots = int(nslots)
        self.launch, self.wait, self.pack = launch, wait, pack
        try:  # launch(k, slot) of rounds 1-4 is still accepted
            self._launch_takes_buf = len(inspect.signature(launch).parameters) >= 3
        except (TypeError, ValueError):
            self._launch_takes_buf = False
        self.on_finalised, self.on_gathered = on_finalised, on_gathered
        self.group, self.dst = group, dst
        self.gather_fn = gather_fn
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # gather="force": run the exchange step at world size 1 too (bench.py --force-gather: the first execution of the RCCL
        # backend and of the device-tensor gather on a one-GPU box)
        self.do_gather = bool(gather) and (self.world > 1 or gather == "force")
        self.packed = packed
        self.nbuf = len(packed) if packed is not None else 0
        if self.do_gather and packed is not None and self.nbuf < self.nslots:
            # the gather of step k is issued when step k + nslots is about to be launched; with fewer buffers than steps
            # in flight step k + nbuf would pack into the buffer of step k before that gather has read it
            raise ValueError(f"with the gather on, {self.nslots} steps in flight need at least {self.nslots} packed buffers "
                             f"(nslots + 2 keeps the gather off the launch path), got {self.nbuf}")
        self._gather_marks = []    # per gathered step: (start, end) CUDA events, or a host duration in seconds
        self.pending = []          # (k, timed) enqueued, not yet finalised
        self.inflight = [None] * max(self.nbuf, 1)   # per packed buffer: (handle, k, timed, t0 | start event) of its gather
        self.launch_waited_for_gather = 0            # launches that found their buffer's gather still running
        self.gather_bufs = None
        self.counts = None
        if self.do_gather:
            if pack is None or packed is None or self.nbuf < 2:
                raise ValueError("gathering needs `pack` and at least two packed buffers")
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
                self.gather_bufs = [[torch.empty_like(packed[0]) for _ in range(self.world)] for _ in range(self.nbuf)]

    @staticmethod
    def _handle_done(h):
        """True / False when the handle can tell without blocking, None when it cannot."""
        for name in ("query", "is_completed", "done"):
            f = getattr(h, name, None)
            if callable(f):
                try:
                    return bool(f())
                except Exception:
                    return None
        return None

    def _retire(self, buf):
        """Block until the gather that reads packed[buf] (if any) is complete; then its buffer may be written again."""
        rec = self.inflight[buf]
        if rec is None:
            return
        self.inflight[buf] = None
        h, k, timed, t0 = rec
        if isinstance(h, self._dist.Work):
            h.wait()          # (a c10d Work also has a deprecated .synchronize that does not block)
        else:
            h.synchronize()   # CUDA event, or the handle of a caller-supplied gather_fn
        if timed:
            if self._torch.is_tensor(self.packed[buf]) and self.packed[buf].is_cuda and self.gather_fn is None:
                self._gather_marks.append((t0, h))
            else:  # host clock from issue to the moment the completion was observed (an upper bound)
                import time as _time
                self._gather_marks.append(_time.perf_counter() - t0)
        if self.on_gathered is not None and self.rank == self.dst:
            self.on_gathered(k, [b[:c] for b, c in zip(self.gather_bufs[buf], self.counts)])

    def _finalise(self, k, timed):
        self.wait(k % self.nslots)
        if self.on_finalised is not None:
            self.on_finalised(k, timed)
        if self.do_gather:
            import time as _time

            buf = k % self.nbuf
            src = self.packed[buf]
            dstbufs = self.gather_bufs[buf] if self.rank == self.dst else None
            if self.gather_fn is not None:
                t0 = _time.perf_counter()
                h = self.gather_fn(src, dstbufs)
            elif src.is_cuda:  # the gather runs on torch's stream: guard the buffer's reuse with an event
                t0 = self._torch.cuda.Event(enable_timing=True)
                t0.record()
                self._dist.gather(src, dstbufs, dst=self.dst, group=self.group)
                h = self._torch.cuda.Event(enable_timing=True)
                h.record()
            else:
                t0 = _time.perf_counter()
                h = self._dist.gather(src, dstbufs, dst=self.dst, group=self.group, async_op=True)
            self.inflight[buf] = (h, k, timed, t0)

    def step(self, k, timed=True):
        slot = k % self.nslots
        while len(self.pending) >= self.nslots:  # slot (and its output buffers) of step k - nslots must be free
            self._finalise(*self.pending.pop(0))
        buf = None
        if self.do_gather:
            # the packed buffer of this step must be free BEFORE the decode is enqueued: where the kernels write packed rows
            # themselves (bposd_decode_batch_device_packed) the launch, not the pack step, fills it.  With nbuf = nslots + 2
            # the gather in question was issued two steps ago.
            buf = k % self.nbuf
            if self.inflight[buf] is not None:
                if self._handle_done(self.inflight[buf][0]) is False:
                    self.launch_waited_for_gather += 1
                self._retire(buf)
        if self._launch_takes_buf:
            self.launch(k, slot, buf)
        else:
            self.launch(k, slot)
        if self.do_gather:
            self.pack(slot, buf)
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
        # gathers in step order (on_gathered sees the steps in order)
        for rec_buf in sorted((b for b in range(len(self.inflight)) if self.inflight[b] is not None), key=lambda b: self.inflight[b][1]):
            self._retire(rec_buf)

    def fence(self):
        """Everything enqueued so far is complete on every rank (the bracket of a timed region)."""
        self.drain()
        if self.world > 1:
            self._dist.barrier(group=self.group)
        if self._torch.cuda.is_available() and self._torch.cuda.is_initialized():
            self._torch.cuda.synchronize()
