"""Large-code regime probe: decode a batch of L29k syndromes, print phase timings and invariants."""
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the decoder: see INTEGRATION.md)

from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import l29k

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
q = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
method = sys.argv[3] if len(sys.argv) > 3 else "osd_e"
order = int(sys.argv[4]) if len(sys.argv) > 4 else 15
max_iter = int(sys.argv[5]) if len(sys.argv) > 5 else 100
t0 = time.time()
H = l29k().hz
print("code", H.shape, "nnz", H.nnz, "build %.1fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(0)
err = (rng.random((B, H.shape[1])) < q).astype(np.uint8)
syn = np.ascontiguousarray((H.astype(np.int32) @ err.T.astype(np.int32) % 2).T.astype(np.uint8))
t0 = time.time()
dec = BpOsdDecoder(H, error_rate=q, max_iter=max_iter, bp_method="ms", ms_scaling_factor=0.625,
                   osd_method=method, osd_order=order)
print("ctor %.2fs rank %d" % (time.time() - t0, dec.rank), flush=True)
for rep in range(2):
    t0 = time.time()
    out = dec.decode_batch(syn, want_osd0=True)
    dt = time.time() - t0
    tm = dec.last_timing()
    print("decode %.3fs  timing %s" % (dt, tm), flush=True)
conv = dec.batch_converge
print("converged %.4f  mean iters %.1f  osd %d" % (conv.mean(), dec.batch_iter.mean(), (~conv).sum()))
ok = ((H.astype(np.int32) @ out.T.astype(np.int32)) % 2 == syn.T).all()
ok0 = ((H.astype(np.int32) @ dec.batch_osd0.T.astype(np.int32)) % 2 == syn.T).all()
print("syndrome satisfied: osdw", bool(ok), "osd0", bool(ok0))
w0 = dec.batch_osd0.sum(1)
ww = out.sum(1)
print("weights osd0 mean %.1f  osdw mean %.1f  (osdw <= osd0: %s)  error weight mean %.1f" % (
    w0.mean(), ww.mean(), bool((ww <= w0).all()), err.sum(1).mean()))
