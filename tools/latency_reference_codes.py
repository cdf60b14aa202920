import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import bposd_decoder
from bp_osd_amd.codes import hgp
import glob
base = os.path.join(os.getcwd(), "tests", "golden")
names = sorted(glob.glob(os.path.join(base, "mkmn_*.txt")))
for f in names:
    code = hgp(np.loadtxt(f, dtype=int).astype(np.uint8), compute_logicals=False)
    H = code.hz; m, n = H.shape; q = 0.05
    dec = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
    rng = np.random.default_rng(0)
    errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
    for s in syns[:20]: dec.decode(s)
    t0 = time.perf_counter(); its = []; nc = 0
    for s in syns: dec.decode(s); its.append(dec.iter); nc += (not dec.converge)
    print(os.path.basename(f), H.shape, "decode() %.0f us per call" % ((time.perf_counter() - t0) / 300 * 1e6), "mean iters %.1f" % np.mean(its), "max", max(its), "osd", nc, flush=True)
