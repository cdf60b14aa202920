"""Which outputs of a one-syndrome bposd_decode_batch call cost what (C-ABI, H1922, 300 seeded syndromes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import h1922

code = h1922(compute_logicals=False); H = code.hz; m, n = H.shape; q = 0.05
dec = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
rng = np.random.default_rng(0)
errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
lib, hnd = dec._lib, dec._h
osdw = np.empty(n, np.uint8); osd0 = np.empty(n, np.uint8); bp = np.empty(n, np.uint8); conv = np.empty(1, np.uint8); it = np.empty(1, np.int32); llr = np.empty(n, np.float64)
P = lambda a: a.ctypes.data
for label, args in (("osdw conv iters", (P(osdw), None, None, P(conv), P(it), None)),
                    ("+ osd0 bp", (P(osdw), P(osd0), P(bp), P(conv), P(it), None)),
                    ("+ llr", (P(osdw), None, None, P(conv), P(it), P(llr))),
                    ("all", (P(osdw), P(osd0), P(bp), P(conv), P(it), P(llr)))):
    for s in syns[:20]: lib.bposd_decode_batch(hnd, s.ctypes.data, 1, *args)
    t0 = time.perf_counter()
    for s in syns: lib.bposd_decode_batch(hnd, s.ctypes.data, 1, *args)
    print(f"{label:18s} {(time.perf_counter() - t0) / len(syns) * 1e6:6.1f} us per call", flush=True)
