"""Single-shot decode() latency through the Python class (what a drop-in user of the reference API sees), next to the
C-ABI call alone and to the CPU oracle's per-syndrome time on the same syndromes."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import h1922, surface13
from oracle import OracleDecoder

for name, code, q in (("S13", surface13(), 0.05), ("H1922", h1922(compute_logicals=False), 0.05)):
    H = code.hz; m, n = H.shape
    kw = dict(error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
    dec = bposd_decoder(H, **kw)
    rng = np.random.default_rng(0)
    errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
    for s in syns[:20]: dec.decode(s)
    t0 = time.perf_counter(); its = []
    for s in syns: dec.decode(s); its.append(dec.iter)
    dt = (time.perf_counter() - t0) / len(syns)
    # the C-ABI alone (osdw + converged + iters), same syndromes
    lib, hnd = dec._lib, dec._h
    osdw = np.empty(n, np.uint8); conv = np.empty(1, np.uint8); it = np.empty(1, np.int32)
    t0 = time.perf_counter()
    for s in syns: lib.bposd_decode_batch(hnd, s.ctypes.data, 1, osdw.ctypes.data, None, None, conv.ctypes.data, it.ctypes.data, None)
    dc = (time.perf_counter() - t0) / len(syns)
    dec.decode_batch(syns)  # (first call of a size class: workspaces are allocated)
    t0 = time.perf_counter(); dec.decode_batch(syns); db = time.perf_counter() - t0
    orc = OracleDecoder(H, **kw)
    t0 = time.perf_counter(); orc.decode_batch(syns, want_llr=False); do = (time.perf_counter() - t0) / len(syns)
    print(f"{name}: decode() {dt*1e6:.0f} us per call (mean {np.mean(its):.0f} BP iterations; max {np.max(its)}); C-ABI alone {dc*1e6:.0f} us; "
          f"decode_batch(300), warm, {db*1e3:.2f} ms total; CPU oracle {do*1e6:.0f} us per syndrome", flush=True)
