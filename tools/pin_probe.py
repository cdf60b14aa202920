"""Where does the host-pointer path spend its time?  C-ABI with warm buffers vs decode_batch (buffer pool).
Page-locked buffers (hipHostMalloc: 41 ms per 252 MB) were measured in round 1 and gained nothing: 43.5 ms either way."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922
import ctypes as C
H = h1922(compute_logicals=False).hz; B = 131072
_, syn = make_batch(H, 0.05, B, seed=0)
dec = BpOsdDecoder(H, error_rate=0.05, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
lib = dec._lib
page = np.empty((B, 1922), np.uint8)
conv = np.empty(B, np.uint8); iters = np.empty(B, np.int32)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    lib.bposd_decode_batch(dec._h, syn.ctypes.data, B, page.ctypes.data, None, None, conv.ctypes.data, iters.ctypes.data, None)
    ts.append(time.perf_counter() - t0)
print("C-ABI, preallocated pageable buffers: %s ms" % " ".join("%.1f" % (t * 1e3) for t in ts))
for _ in range(3):
    t0 = time.perf_counter(); out = dec.decode_batch(syn, want_osd0=False, want_bp=False); t1 = time.perf_counter()
    print("decode_batch (pool): %.1f ms, pool %d" % ((t1 - t0) * 1e3, len(dec.__dict__.get("_out_pool", []))))
    del out
