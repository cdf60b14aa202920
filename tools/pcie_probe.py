"""PCIe-inclusive throughput of the host-pointer API (bposd_decode_batch): numpy in, numpy out."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922
H = h1922(compute_logicals=False).hz; B = 131072
_, syn = make_batch(H, 0.05, B, seed=0)
dec = BpOsdDecoder(H, error_rate=0.05, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
dec.decode_batch(syn[:4096], want_osd0=False, want_bp=False)
for label, kw in (("osdw only", dict(want_osd0=False, want_bp=False)), ("osdw+osd0+bp", dict())):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); dec.decode_batch(syn, **kw); ts.append(time.perf_counter() - t0)
    t = min(ts); k = dec.last_timing()
    print(f"{label}: {B / t / 1e6:.2f} M syndromes/s host-to-host ({t*1e3:.1f} ms; kernels {k['bp_ms'] + k['osd_ms']:.1f} ms)")
