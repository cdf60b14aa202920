"""BP alone on the 14520 x 29524 code (BASELINE configs[4] without OSD): kernel time per 1024 syndromes.
usage: [BPOSD_LARGE_WG_CAP=k] python tools/bp_large_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import l29k
H = l29k().hz
m, n = H.shape
B, q = 1024, 0.05
rng = np.random.default_rng(0)
err = (rng.random((B, n)) < q).astype(np.uint8)
syn = np.ascontiguousarray((np.asarray(H @ err.T) % 2).T.astype(np.uint8))
dec = BpOsdDecoder(H, error_rate=q, max_iter=100, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_off")
for rep in range(3):
    dec.decode_batch(syn, want_osd0=False, want_bp=False)
    t = dec.last_timing()
    its = t["bp_iterations"]
    print(f"cap={os.environ.get('BPOSD_LARGE_WG_CAP', '-')} bp_ms={t['bp_ms']:.2f} iterations={its} -> {its * (4 * H.nnz + 2 * n) * 8 / t['bp_ms'] / 1e9:.2f} TB/s algorithmic", flush=True)
