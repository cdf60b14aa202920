"""Host-to-host decode_batch_into steps for a timeline trace (rocprofv3 --kernel-trace --memory-copy-trace)."""
import sys, time
import numpy as np
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922
sys.path.insert(0, ".")
from bench import make_batch
H = h1922(compute_logicals=False).hz
m, n = H.shape
B = 131072
dec = BpOsdDecoder(H, error_rate=0.05, max_iter=0, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=7)
err, syn = make_batch(H, 0.05, B, seed=0)
h_syn = dec.pinned_empty((B, m)); h_syn[:] = syn
osdw = dec.pinned_empty((B, n)); conv = dec.pinned_empty((B,)); iters = dec.pinned_empty((B,), np.int32)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t = time.perf_counter()
    dec.decode_batch_into(h_syn, osdw, converged=conv, iters=iters)
    print("step %d: %.2f ms" % (k, 1e3 * (time.perf_counter() - t)), flush=True)
