"""Single-shot decode() latency through the Python class (what a drop-in user of the reference API sees)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import h1922, surface13
for name, code, q in (("S13", surface13(), 0.05), ("H1922", h1922(compute_logicals=False), 0.05)):
    H = code.hz; n = H.shape[1]
    dec = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
    rng = np.random.default_rng(0)
    errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.asarray((H @ errs.T) % 2).T.astype(np.uint8)
    for s in syns[:20]: dec.decode(s)
    t0 = time.perf_counter()
    its = []
    for s in syns: dec.decode(s); its.append(dec.iter)
    dt = (time.perf_counter() - t0) / len(syns)
    t0 = time.perf_counter(); dec.decode_batch(syns); db = time.perf_counter() - t0
    print(f"{name}: decode() {dt*1e6:.0f} us per call (mean {np.mean(its):.0f} BP iterations); decode_batch(300) {db*1e3:.2f} ms total")
