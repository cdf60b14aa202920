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
