"""N > 1 path on CPU: world_size-2 gloo run of the shard -> decode -> gather pipeline.  The local
decoder is the CPU oracle (stand-in for the per-GPU handle; tests may use it as the checker)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bp_osd_amd.sharding import shard_bounds


def test_shard_bounds_partition():
    for total in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bp_osd_amd.codes import surface13
        from bp_osd_amd.sharding import decode_sharded, reduce_counts
        from oracle import OracleDecoder

        code = surface13()
        rng = np.random.default_rng(0)  # same global batch on every rank
        err = (rng.random((total, 13)) < 0.1).astype(np.uint8)
        syn = (err @ code.hz.toarray().T % 2).astype(np.uint8)
        dec = OracleDecoder(code.hz, error_rate=0.1, max_iter=3, bp_method="ms", ms_scaling_factor=0,
                            osd_method="osd_cs", osd_order=4)
        seen = []

        def local(shard):
            seen.append(len(shard))
            return dec.decode_batch(shard, want_llr=False)["osdw"] if len(shard) else np.zeros((0, 13), np.uint8)

        out = decode_sharded(local, syn)
        counts = reduce_counts([seen[0], 1])
        assert counts == [total, world]
        if rank == 0:
            full = dec.decode_batch(syn, want_llr=False)["osdw"]
            assert out.shape == (total, 13)
            assert (out.numpy() == full).all()
            q_out.put("ok")
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [33, 64])
def test_world_size_2_gloo_shard_decode_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


# ---------------------------------------------------------------------------------------------------------------
# StepPipeline: the loop bench.py --gpus N runs (decode -> pack -> gather, two slots, double-buffered packed rows),
# driven here on gloo with a stub decoder whose "stream" is a per-slot queue executed in order at wait() time.
def _pack_words(bits):
    """numpy twin of bposd_pack_rows_device: bit (i & 63) of word (i >> 6) = bits[:, i]."""
    B, n = bits.shape
    wpr = (n + 63) // 64
    padded = np.zeros((B, wpr * 64), dtype=np.uint8)
    padded[:, :n] = bits
    return np.packbits(padded, axis=1, bitorder="little").view(np.int64).reshape(B, wpr)


def _pipeline_worker(rank, world, port, total, steps, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bp_osd_amd.sharding import StepPipeline, shard_bounds

        n, nslots = 100, 2
        wpr = (n + 63) // 64
        lo, hi = shard_bounds(total, rank, world)
        rows = hi - lo
        rows_max = max(shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world))

        def truth(k, r):  # the "corrections" of rank r at step k: a deterministic function of (step, global row, column)
            a, b = shard_bounds(total, r, world)
            g = np.arange(a, b)[:, None]
            return ((g * 31 + np.arange(n)[None, :] * 17 + k * 7) % 5 == 0).astype(np.uint8)

        out = [np.zeros((rows, n), np.uint8) for _ in range(nslots)]       # a slot's output buffer
        packed = [torch.zeros((rows_max, wpr), dtype=torch.int64) for _ in range(2)]
        queues = [[] for _ in range(nslots)]                                # stream-ordered work of each slot
        log = []

        def launch(k, slot):
            queues[slot].append(lambda: out[slot].__setitem__(slice(None), truth(k, rank)))

        def pack(slot, buf):
            queues[slot].append(lambda: packed[buf][:rows].copy_(torch.from_numpy(_pack_words(out[slot]))))

        def wait(slot):
            for job in queues[slot]:
                job()
            queues[slot].clear()

        checked = []

        def on_gathered(k, shards):
            assert [s.shape[0] for s in shards] == [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0]
                                                    for r in range(world)]
            for r, s in enumerate(shards):
                assert (s.numpy() == _pack_words(truth(k, r))).all(), (k, r)
            checked.append(k)

        pipe = StepPipeline(nslots, launch, wait, pack=pack, packed=packed, rows=rows,
                            on_finalised=lambda k, timed: log.append((k, timed)), on_gathered=on_gathered)
        pipe.step(0, False)
        pipe.fence()
        for k in range(1, steps + 1):
            pipe.step(k, True)
            assert len(pipe.pending) <= nslots
        pipe.fence()
        assert log == [(0, False)] + [(k, True) for k in range(1, steps + 1)]
        assert len(pipe.gather_ms()) == steps  # one gather duration per timed step
        # three steps in flight on TWO packed buffers would repack a buffer before its gather has been issued: refused
        try:
            StepPipeline(3, launch, wait, pack=pack, packed=packed, rows=rows)
            raise AssertionError("nslots = 3 with two packed buffers must be refused")
        except ValueError:
            pass
        if rank == 0:
            assert checked == list(range(steps + 1))
            q_out.put("ok")
        else:
            assert checked == []
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [33, 64])
def test_world_size_2_gloo_step_pipeline(total):
    """Buffer reuse over 6 steps (3 rounds of each slot and of each packed buffer), unequal shards (17 + 16 of 33)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, total, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def _slow_gather_worker(rank, world, port, nbuf, q_out):
    """Two steps in flight, a gather that takes 0.15 s and a decode that takes 0.2 s: with nslots + 2 packed buffers the launch of
    step k + 2 is enqueued while gather k is still running and no launch ever waits for a gather; with the pair of
    rounds 1-4 every launch waits for the gather issued just before it."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bp_osd_amd.sharding import StepPipeline

        rows, wpr, nslots, steps = 8, 2, 2, 7
        packed = [torch.zeros((rows, wpr), dtype=torch.int64) for _ in range(nbuf)]
        launch_t, done_t, seen = {}, {}, []
        pool = ThreadPoolExecutor(max_workers=1)  # ONE worker: the collectives keep their order on every rank
        kq = []

        class Handle:
            def __init__(self, fut):
                self.fut = fut

            def query(self):
                return self.fut.done()

            def synchronize(self):
                self.fut.result()

        def gather_fn(src, bufs):
            k = kq.pop(0)

            def job():
                time.sleep(0.15)
                dist.gather(src, bufs, dst=0)
                done_t[k] = time.perf_counter()

            return Handle(pool.submit(job))

        def launch(k, slot, buf):
            launch_t[k] = time.perf_counter()
            packed[buf].fill_(1000 * k + rank)  # the launch writes the packed rows itself (native packed decode)

        def wait(slot):
            time.sleep(0.2)

        def on_finalised(k, timed):
            kq.append(k)

        def on_gathered(k, shards):
            seen.append(k)
            for r, s in enumerate(shards):
                assert (s == 1000 * k + r).all(), (k, r)  # no buffer was overwritten before its gather had read it

        pipe = StepPipeline(nslots, launch, wait, pack=lambda slot, buf: None, packed=packed, rows=rows,
                            on_finalised=on_finalised, on_gathered=on_gathered, gather_fn=gather_fn)
        for k in range(steps):
            pipe.step(k, True)
        pipe.fence()
        pool.shutdown()
        assert len(pipe.gather_ms()) == steps
        if rank == 0:
            assert seen == list(range(steps))
        early = [launch_t[k + nslots] < done_t[k] for k in range(steps - nslots)]
        q_out.put((rank, nbuf, pipe.launch_waited_for_gather, early))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nbuf", [4, 2])
def test_gather_is_off_the_launch_path_with_nslots_plus_two_buffers(nbuf):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slow_gather_worker, args=(r, 2, port, nbuf, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for _ in range(2):
        rank, nb, waited, early = q.get(timeout=5)
        if nb == 4:  # step k + 2 was launched while gather k was still running, and never had to wait for a buffer
            assert waited == 0 and all(early), (rank, waited, early)
        else:        # the double buffer of rounds 1-4: every launch from step 2 on waits for the gather issued just before it
            assert waited >= 4 and not any(early), (rank, waited, early)


def test_step_pipeline_single_process_needs_no_process_group():
    from bp_osd_amd.sharding import StepPipeline

    done = []
    pipe = StepPipeline(2, launch=lambda k, slot: done.append(("launch", k, slot)), wait=lambda slot: done.append(("wait", slot)))
    for k in range(3):
        pipe.step(k)
    pipe.fence()
    assert [d for d in done if d[0] == "launch"] == [("launch", 0, 0), ("launch", 1, 1), ("launch", 2, 0)]
    assert done.index(("wait", 0)) < done.index(("launch", 2, 0))  # slot 0 is reused only after step 0 was finalised
