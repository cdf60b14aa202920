"""OSD kernel timing probe: H1922, noisy syndromes, BP capped so that every shot goes through OSD."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922
code = h1922(compute_logicals=False); H = code.hz
rng = np.random.default_rng(0); q = 0.08; B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
err = (rng.random((B, 1922)) < q).astype(np.uint8); syn = np.asarray((H @ err.T) % 2).T.astype(np.uint8)
for osd in (("osd0", 0), ("osd_cs", 7), ("osd_cs", 60), ("osd_e", 10)):
    dec = BpOsdDecoder(H, error_rate=q, max_iter=3, bp_method="ms", ms_scaling_factor=0, osd_method=osd[0], osd_order=osd[1])
    dec.decode_batch(syn[:256]); 
    dec.decode_batch(syn)
    t = dec.last_timing()
    per = t["osd_ms"] / max(1, -(-t["osd_invocations"] // 256))
    print(f"{osd}: osd {t['osd_ms']:.2f} ms for {t['osd_invocations']} invocations -> {per:.3f} ms per syndrome per workgroup", flush=True)
