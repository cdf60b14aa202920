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
    rng = np.random.default_rng(0)
    errs = (rng.random((300, n)) < q).astype(np.uint8); syns = np.ascontiguousarray(np.asarray((H @ errs.T) % 2).T.astype(np.uint8))
    for osd_variant in (2, 1):  # 2 = one wave per elimination, 1 = one workgroup per elimination (what auto takes for small calls)
        dec = bposd_decoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=int(os.environ.get("OSD_ORDER", "7")))
        dec.set_osd_variant(osd_variant)
        for s in syns[:20]: dec.decode(s)
        t0 = time.perf_counter(); its = []; nc = 0; t_osd = 0.0
        for s in syns:
            t1 = time.perf_counter(); dec.decode(s); dt = time.perf_counter() - t1
            its.append(dec.iter); nc += (not dec.converge); t_osd += dt if not dec.converge else 0.0
        tot = time.perf_counter() - t0
        print(os.path.basename(f), H.shape, "osd kernel", dec.last_osd_kernel(), "decode() %.0f us per call" % (tot / 300 * 1e6), "mean iters %.1f" % np.mean(its), "max", max(its),
              "osd", nc, "-> %.0f us per call that needed OSD, %.0f us otherwise" % (t_osd / max(nc, 1) * 1e6, (tot - t_osd) / max(300 - nc, 1) * 1e6), flush=True)
