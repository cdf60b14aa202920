"""Which L29k eliminations are slow: the non-converged syndromes of a batch decoded one at a time (OSD-E 15)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import l29k
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
q = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
code = l29k(compute_logicals=False); H = code.hz
rng = np.random.default_rng(0)
err = (rng.random((B, H.shape[1])) < q).astype(np.uint8); syn = np.ascontiguousarray(np.asarray((H @ err.T) % 2).T.astype(np.uint8))
dec = BpOsdDecoder(H, error_rate=q, max_iter=100, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_e", osd_order=15)
dec.decode_batch(syn); bad = np.where(~dec.batch_converge)[0]
print("non-converged", len(bad), "batch osd_ms", dec.last_timing()["osd_ms"], flush=True)
ts = []
for i in bad[:40]:
    dec.decode_batch(syn[i:i + 1]); t0 = time.perf_counter(); dec.decode_batch(syn[i:i + 1]); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts); print("per-syndrome decode ms: min %.1f median %.1f max %.1f" % (ts.min(), np.median(ts), ts.max()), " slowest index", int(bad[np.argmax(ts)]), flush=True)
if os.environ.get("BPOSD_OSD_DEBUG"):
    i = int(bad[np.argmax(ts)]); print("slowest again (diag):", flush=True); dec.decode_batch(syn[i:i + 1])
    i = int(bad[np.argmin(ts)]); print("fastest (diag):", flush=True); dec.decode_batch(syn[i:i + 1])
