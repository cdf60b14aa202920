import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import hgp
H = hgp(np.loadtxt("tests/golden/mkmn_16_4_6.txt", dtype=int).astype(np.uint8), compute_logicals=False).hz
m, n = H.shape; q = 0.03
rng = np.random.default_rng(0)
errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
for v in (0, 1, 2):
    for osd in ("osd_cs", "osd0"):
        dec = bposd_decoder(H, error_rate=q, max_iter=50, bp_method="ms", ms_scaling_factor=0, osd_method=osd, osd_order=7 if osd == "osd_cs" else 0)
        dec.set_bp_variant(v)
        for s in syns[:20]: dec.decode(s)
        t0 = time.perf_counter(); its = []; nc = 0
        for s in syns: dec.decode(s); its.append(dec.iter); nc += (not dec.converge)
        print("variant", v, osd, "decode() %.0f us" % ((time.perf_counter() - t0) / 300 * 1e6), "mean iters %.1f" % np.mean(its), "osd", nc, flush=True)
