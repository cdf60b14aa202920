"""Batch size at which the throughput variant of the local-edge BP kernel overtakes the latency variant (H1922)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import h1922

code = h1922(compute_logicals=False); H = code.hz; m, n = H.shape; q = 0.05
rng = np.random.default_rng(0)
errs = (rng.random((131072, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
decs = {}
for v in (22, 26):
    decs[v] = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
    decs[v].set_bp_variant(v)
for B in [int(x) for x in (sys.argv[1:] or [128, 256, 512, 768, 1024, 2048, 4096, 8192])]:
    row = []
    for v in (22, 26):
        d = decs[v]; d.decode_batch(syns[:B]); ts = []
        for r in range(min(4, 131072 // B)):
            t0 = time.perf_counter(); d.decode_batch(syns[r * B:(r + 1) * B]); ts.append(time.perf_counter() - t0)
        row.append(np.mean(ts) * 1e3)
    print(f"B {B:5d}: throughput variant {row[0]:7.3f} ms   latency variant {row[1]:7.3f} ms", flush=True)
