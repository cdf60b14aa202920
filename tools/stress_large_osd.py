"""One-off stress of the HBM-resident OSD kernel against the oracle: 2025 x 4050 code, several noise levels / BP depths /
OSD settings, every output compared bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import circulant, hgp
from oracle import OracleDecoder
H = hgp(circulant(45, (0, 2, 5)), compute_logicals=False).hz
n = H.shape[1]; bad = 0; total = 0
for seed, (q, mi, method, order, tie) in enumerate([(0.04, 3, "osd_e", 6, 0), (0.06, 1, "osd_0", 0, 1), (0.08, 8, "osd_e", 10, 0), (0.03, 20, "osd_cs", 4, 0),
                                                    (0.10, 2, "osd_e", 3, 1), (0.05, 5, "osd_0", 0, 0)]):
    rng = np.random.default_rng(100 + seed)
    err = (rng.random((160, n)) < q).astype(np.uint8); syn = np.ascontiguousarray(np.asarray((H @ err.T) % 2).T.astype(np.uint8))
    kw = dict(error_rate=q, max_iter=mi, bp_method="ms", ms_scaling_factor=0.7, osd_method=method, osd_order=order, sort_tie_policy=tie)
    g = BpOsdDecoder(H, **kw); got = g.decode_batch(syn, want_osd0=True)
    t0 = time.time(); ref = OracleDecoder(H, **kw).decode_batch(syn)
    ok = (got == ref["osdw"]).all() and (g.batch_osd0 == ref["osd0"]).all() and (g.batch_iter == ref["iters"]).all()
    nz = int((~g.batch_converge).sum()); total += nz; bad += 0 if ok else 1
    print(kw["osd_method"], order, "q", q, "max_iter", mi, "non-converged", nz, "exact", bool(ok), "(oracle %.1f s)" % (time.time() - t0), flush=True)
print("OK" if bad == 0 else "MISMATCH", total, "eliminations")
