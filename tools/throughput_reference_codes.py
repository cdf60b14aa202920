"""BP cost per edge-iteration on the reference's three example codes (and H1922) per BP kernel variant.
usage: python tools/throughput_reference_codes.py [variant ...]   (0 = auto, 1 = generic LDS kernel, 32 = class kernel)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import hgp, h1922
import glob
variants = [int(v) for v in sys.argv[1:]] or [0]
codes = [(os.path.basename(f), hgp(np.loadtxt(f, dtype=int).astype(np.uint8), compute_logicals=False).hz) for f in sorted(glob.glob("tests/golden/mkmn_*.txt"))]
codes.append(("h1922", h1922(compute_logicals=False).hz))
for name, H in codes:
    m, n = H.shape; q = 0.03; B = 262144 if n < 1000 else 131072
    rng = np.random.default_rng(0)
    errs = (rng.random((B, n)) < q).astype(np.uint8); syns = np.ascontiguousarray((np.asarray(H @ errs.T) % 2).T.astype(np.uint8))
    ref = None
    for v in variants:
        dec = BpOsdDecoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
        try:
            dec.set_bp_variant(v)
        except Exception as e:
            print(name, "variant", v, "unavailable:", e); continue
        out = dec.decode_batch(syns, want_osd0=False, want_bp=False); out = dec.decode_batch(syns, want_osd0=False, want_bp=False).copy()
        same = "" if ref is None else (" outputs equal first variant: %s" % bool((out == ref).all()))
        if ref is None: ref = out
        t = dec.last_timing(); E = H.nnz
        print(name, H.shape, "variant", v, "nnz", E, "B", B, "bp_ms %.2f osd_ms %.2f" % (t["bp_ms"], t["osd_ms"]), "iters/syn %.1f" % (t["bp_iterations"] / B), "osd", t["osd_invocations"],
              "-> %.3f ns per syndrome-iteration, %.2f ps per edge-iteration" % (t["bp_ms"] * 1e6 / t["bp_iterations"], t["bp_ms"] * 1e9 / t["bp_iterations"] / E) + same, flush=True)
