"""Where a 300-syndrome decode_batch spends its time (H1922)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import h1922
code = h1922(compute_logicals=False); H = code.hz; m, n = H.shape; q = 0.05
rng = np.random.default_rng(0)
errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
dec = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
for B in (100, 300):
    for rep in range(3):
        t0 = time.perf_counter(); dec.decode_batch(syns[:B]); dt = time.perf_counter() - t0
        print(B, f"{dt*1e3:.3f} ms", dec.last_timing(), "max iters", dec.batch_iter.max(), "nonconv", int((~dec.batch_converge).sum()), flush=True)
