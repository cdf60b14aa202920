"""One-syndrome decode() latency of the local-edge BP kernel variants (H1922, 300 seeded syndromes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import h1922

code = h1922(compute_logicals=False); H = code.hz; m, n = H.shape; q = 0.05
rng = np.random.default_rng(0)
errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
for v in [int(x) for x in (sys.argv[1:] or ["0", "26", "18", "24", "2"])]:
    dec = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
    dec.set_bp_variant(v)
    for s in syns[:20]: dec.decode(s)
    t0 = time.perf_counter(); its = 0
    for s in syns: dec.decode(s); its += dec.iter
    dt = (time.perf_counter() - t0) / len(syns)
    for B in (16, 256):
        dec.decode_batch(syns[:B]); t0 = time.perf_counter()
        for _ in range(5): dec.decode_batch(syns[:B])
        print(f"variant {v}: decode_batch({B}) {(time.perf_counter() - t0) / 5 * 1e6:.0f} us", flush=True)
    print(f"variant {v}: decode() {dt * 1e6:.0f} us per call, {its / len(syns):.0f} iterations mean", flush=True)
