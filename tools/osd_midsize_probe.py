"""OSD kernels on mid-size codes (320 < checks <= 1024): osd_kernel (one workgroup per elimination, variant 1) against
osd_mw_kernel (a few waves per elimination, rows in registers, variant 2) -- kernel time per batch, outputs compared.
usage: python tools/osd_midsize_probe.py [case ...]   cases: hgp900 surface21 surface25 surface19 h1922"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import hgp, rep_code, h1922
ROOT = os.getcwd()
def code(name):
    if name == "hgp900":
        seed = np.loadtxt(os.path.join(ROOT, "tests", "golden", "mkmn_24_6_10.txt"), dtype=int).astype(np.uint8)
        return hgp(seed, compute_logicals=False).hz, 65536, 0.05, dict(max_iter=0, osd_method="osd_cs", osd_order=42)
    if name.startswith("surface"):
        d = int(name[7:])
        return hgp(rep_code(d), compute_logicals=False).hz, 32768, 0.05, dict(max_iter=30, osd_method="osd_cs", osd_order=10)
    if name == "h1922":
        return h1922(compute_logicals=False).hz, 4096, 0.085, dict(max_iter=12, osd_method="osd_cs", osd_order=7)
    raise SystemExit(name)
for name in (sys.argv[1:] or ["hgp900", "surface21", "surface25", "h1922"]):
    H, B, q, kw = code(name)
    m, n = H.shape
    rng = np.random.default_rng(1)
    err = (rng.random((B, n)) < q).astype(np.uint8)
    syn = np.ascontiguousarray((np.asarray(H @ err.T) % 2).T.astype(np.uint8))
    ref = None
    for v in (1, 2):
        dec = BpOsdDecoder(H, error_rate=q, bp_method="ms", ms_scaling_factor=0, **kw)
        dec.set_osd_variant(v)
        dec.decode_batch(syn)
        out = dec.decode_batch(syn, want_osd0=True).copy(); o0 = dec.batch_osd0.copy()
        t = dec.last_timing()
        same = "" if ref is None else " identical to variant 1: %s" % bool((out == ref[0]).all() and (o0 == ref[1]).all())
        if ref is None: ref = (out, o0)
        print(f"{name} {m}x{n} {kw['osd_method']} {kw['osd_order']} variant {v} {dec.last_osd_kernel()}: osd {t['osd_ms']:.2f} ms for {t['osd_invocations']} eliminations "
              f"({1e3 * t['osd_ms'] / max(t['osd_invocations'], 1):.2f} us each), bp {t['bp_ms']:.2f} ms" + same, flush=True)
