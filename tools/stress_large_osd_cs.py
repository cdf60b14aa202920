import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import circulant, hgp
from oracle import OracleDecoder
H = hgp(circulant(45, (0, 2, 5)), compute_logicals=False).hz
n = H.shape[1]; bad = 0
for seed, (q, mi, order, tie, nonuni) in enumerate([(0.06, 2, 5, 0, False), (0.09, 1, 8, 1, False), (0.04, 6, 3, 0, True), (0.02, 30, 10, 0, False)]):
    rng = np.random.default_rng(500 + seed)
    err = (rng.random((128, n)) < q).astype(np.uint8); syn = np.ascontiguousarray(np.asarray((H @ err.T) % 2).T.astype(np.uint8))
    probs = rng.uniform(0.01, 0.2, size=n) if nonuni else np.full(n, q)
    kw = dict(channel_probs=probs, max_iter=mi, bp_method="ms", ms_scaling_factor=0.7, osd_method="osd_cs", osd_order=order, sort_tie_policy=tie)
    g = BpOsdDecoder(H, **kw); got = g.decode_batch(syn, want_osd0=True)
    ref = OracleDecoder(H, **kw).decode_batch(syn)
    ok = (got == ref["osdw"]).all() and (g.batch_osd0 == ref["osd0"]).all()
    bad += 0 if ok else 1
    print("osd_cs", order, "q", q, "max_iter", mi, "nonuniform", nonuni, "non-converged", int((~g.batch_converge).sum()), "exact", bool(ok), flush=True)
print("OK" if bad == 0 else "MISMATCH")
