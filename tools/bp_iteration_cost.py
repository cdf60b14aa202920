"""Pure BP cost per iteration and per syndrome: syndromes that never converge (q = 0.25, OSD off) at two iteration caps.
usage: python tools/bp_iteration_cost.py [variant ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import hgp, h1922
import glob
variants = [int(v) for v in sys.argv[1:]] or [0]
method = os.environ.get("BP_METHOD", "ms")
codes = [(os.path.basename(f), hgp(np.loadtxt(f, dtype=int).astype(np.uint8), compute_logicals=False).hz) for f in sorted(glob.glob("tests/golden/mkmn_*.txt"))]
codes.append(("h1922", h1922(compute_logicals=False).hz))
# EXTRA_CODES=surface:15,toric:12 adds hypergraph products of the distance-d repetition / ring code
if os.environ.get("EXTRA_CODES"):
    from bp_osd_amd.codes import rep_code, ring_code
    extra = []
    for spec in os.environ["EXTRA_CODES"].split(","):
        kind, d = spec.split(":")
        seed = rep_code(int(d)) if kind == "surface" else ring_code(int(d))
        extra.append((spec, hgp(seed, compute_logicals=False).hz))
    codes = extra if os.environ.get("ONLY_EXTRA") else codes + extra
for name, H in codes:
    m, n = H.shape; q = 0.25; B = 65536
    rng = np.random.default_rng(0)
    errs = (rng.random((B, n)) < q).astype(np.uint8); syns = np.ascontiguousarray((np.asarray(H @ errs.T) % 2).T.astype(np.uint8))
    for v in variants:
        res = {}
        for cap in (16, 48):
            dec = BpOsdDecoder(H, error_rate=0.05, max_iter=cap, bp_method=method, ms_scaling_factor=0, osd_method="osd_off")
            try:
                dec.set_bp_variant(v)
            except Exception as e:
                print(name, "variant", v, "unavailable:", e); res = None; break
            dec.decode_batch(syns, want_osd0=False, want_bp=False); dec.decode_batch(syns, want_osd0=False, want_bp=False)
            t = dec.last_timing()
            res[cap] = (t["bp_ms"], t["bp_iterations"] / B)
        if not res: continue
        (t1, i1), (t2, i2) = res[16], res[48]
        per_it = (t2 - t1) * 1e6 / ((i2 - i1) * B)
        per_syn = t1 * 1e6 / B - per_it * i1
        print(f"{name} variant {v} {method}: {per_it:.3f} ns per syndrome-iteration = {per_it * 1e3 / H.nnz:.3f} ps per edge-iteration, {per_syn:.1f} ns per syndrome outside the iterations "
              f"(= {per_syn / per_it:.1f} iterations); caps 16 / 48: {t1:.2f} / {t2:.2f} ms, mean iterations {i1:.1f} / {i2:.1f}", flush=True)
