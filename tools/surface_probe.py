"""Surface / toric codes end to end (the reference README's example family at useful distances): kernel times per batch.
usage: python tools/surface_probe.py [d ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import hgp, rep_code, ring_code
B = int(os.environ.get("B", 65536))
q = float(os.environ.get("Q", 0.05))
for spec in (sys.argv[1:] or ["surface:9", "surface:13", "surface:17", "surface:21", "toric:12", "toric:20"]):
    kind, d = spec.split(":")
    H = hgp(rep_code(int(d)) if kind == "surface" else ring_code(int(d)), compute_logicals=False).hz
    m, n = H.shape
    rng = np.random.default_rng(1)
    err = (rng.random((B, n)) < q).astype(np.uint8)
    syn = np.ascontiguousarray((np.asarray(H @ err.T) % 2).T.astype(np.uint8))
    for osd, order in (("osd_cs", 10), ("osd_e", 8), ("osd0", 0)):
        dec = BpOsdDecoder(H, error_rate=q, max_iter=int(os.environ.get("MAX_ITER", n)), bp_method="ms", ms_scaling_factor=0, osd_method=osd, osd_order=order)
        if os.environ.get("OSD_VARIANT"):
            dec.set_osd_variant(int(os.environ["OSD_VARIANT"]))
        dec.decode_batch(syn); dec.decode_batch(syn)
        t = dec.last_timing()
        print(f"{spec} {m}x{n} q={q} {osd}{order}: bp {t['bp_ms']:.2f} ms ({dec.bp_kernel_info()['kernel']}), osd {t['osd_ms']:.2f} ms "
              f"({dec.last_osd_kernel()}, {t['osd_invocations']} eliminations = {100.0 * t['osd_invocations'] / B:.0f} %), "
              f"mean iterations {t['bp_iterations'] / B:.1f} -> {B / (t['bp_ms'] + t['osd_ms']) / 1e3:.2f} M syndromes/s", flush=True)
