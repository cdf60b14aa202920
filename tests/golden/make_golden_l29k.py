"""Golden vectors for the large configuration (L29k: 14520 x 29524 HGP code), from the CPU oracle.

Same provenance as make_golden.py (this repository's oracle; no reference code).  Kept separate because the
oracle needs minutes here: the elimination of one 14520 x 29524 matrix takes ~4.5 s and every OSD-W candidate
another ~7 ms, so the orders are kept small (osd_e 10 -> 1023 candidates; osd_cs 3 -> 15004 + 3).

    python tests/golden/make_golden_l29k.py            (e10 + cs3)
    python tests/golden/make_golden_l29k.py e15        (BASELINE configs[4]'s own settings: osd_e 15, max_iter 100,
                                                        q = 0.05; 32767 candidates x ~7 ms -> ~4 min per OSD shot)
    python tests/golden/make_golden_l29k.py r4         (round 4: e15b, cs42, cs16sel -- 26 more eliminations, one host
                                                        thread per shot, ~20 min on 6 threads)
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from bp_osd_amd.codes import l29k  # noqa: E402
from oracle import OracleDecoder  # noqa: E402


def pack(a):
    return np.packbits(np.asarray(a, dtype=np.uint8), axis=1)


def make_e15(H, digest):
    """configs[4] at its stated settings.  Shots are picked with a BP-only pass (fast): two that do not converge in
    100 iterations (they go through the 32767-candidate OSD-E sweep) and one that does."""
    q = 0.05
    rng = np.random.default_rng(15)
    err = (rng.random((24, H.shape[1])) < q).astype(np.uint8)
    syn = np.ascontiguousarray((H.astype(np.int32) @ err.T.astype(np.int32) % 2).T.astype(np.uint8))
    bp_only = dict(error_rate=q, max_iter=100, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_off", osd_order=0)
    conv = OracleDecoder(H, **bp_only).decode_batch(syn, want_llr=False)["converged"].astype(bool)
    pick = list(np.flatnonzero(~conv)[:2]) + list(np.flatnonzero(conv)[:1])
    sub = np.ascontiguousarray(syn[pick])
    cfg = dict(error_rate=q, max_iter=100, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_e", osd_order=15)
    t0 = time.time()
    r = OracleDecoder(H, **cfg).decode_batch(sub)
    print("e15", "%.1fs" % (time.time() - t0), "converged", r["converged"], "weights", r["osdw"].sum(1), r["osd0"].sum(1), flush=True)
    np.savez_compressed(os.path.join(HERE, "l29k_golden_e15.npz"), cfg=repr(cfg), code_sha256=digest,
                        syn=pack(sub), osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]),
                        converged=r["converged"], iters=r["iters"])


def _one_shot(args):
    """One shot on an oracle handle of its own (ctypes releases the GIL, so shots run on separate host threads)."""
    H, cfg, syn, probs = args
    o = OracleDecoder(H, **cfg)
    if probs is not None:
        o.update_channel_probs(probs)
    t0 = time.time()
    r = o.decode(syn)
    print("  shot done %.0fs converged %d weights %d %d" % (time.time() - t0, r["converged"], r["osdw"].sum(), r["osd0"].sum()), flush=True)
    return r


def _nonconverged(H, cfg, syn, want_nc, want_c, probs=None):
    """Indices of the first `want_nc` shots BP leaves unconverged and the first `want_c` it converges (BP-only pass)."""
    bp_only = dict(cfg, osd_method="osd_off", osd_order=0)
    o = OracleDecoder(H, **bp_only)
    if probs is None:
        conv = o.decode_batch(syn, want_llr=False)["converged"].astype(bool)
    else:
        conv = np.zeros(len(syn), bool)
        for b in range(len(syn)):
            o.update_channel_probs(probs[b])
            conv[b] = bool(o.decode(syn[b])["converged"])
    return list(np.flatnonzero(~conv)[:want_nc]) + list(np.flatnonzero(conv)[:want_c])


def make_round4(H, digest, threads=6):
    """Round 4: 26 more eliminations on the 16-rows-per-thread instance of the HBM-resident OSD kernel.
    e15b = configs[4]'s settings again (12 through OSD + 1 converged); cs42 = the reference example's OSD setting
    (examples/qldpc_decode_example.py:15-16) with a uniform channel (8 + 1); cs16sel = osd_cs 16 with the per-shot
    two-valued channel of the harness's channel_update (css_decode_sim.py:207-248; 6 + 1)."""
    from concurrent.futures import ThreadPoolExecutor

    q = 0.05
    m, n = H.shape
    base = dict(error_rate=q, max_iter=100, bp_method="ms", ms_scaling_factor=0.625)
    rng = np.random.default_rng(2904)
    err = (rng.random((160, n)) < q).astype(np.uint8)
    syn_all = np.ascontiguousarray((H.astype(np.int32) @ err.T.astype(np.int32) % 2).T.astype(np.uint8))
    jobs = {}
    cfg = dict(base, osd_method="osd_e", osd_order=15)
    pick = _nonconverged(H, cfg, syn_all[:64], 12, 1)
    jobs["e15b"] = (cfg, syn_all[:64][pick], None, None, None)
    cfg = dict(base, osd_method="osd_cs", osd_order=42)
    pick = _nonconverged(H, cfg, syn_all[64:112], 8, 1)
    jobs["cs42"] = (cfg, syn_all[64:112][pick], None, None, None)
    cfg = dict(base, osd_method="osd_cs", osd_order=16)
    sel = (rng.random((48, n)) < 0.08).astype(np.uint8)
    alt_p = 0.3
    probs = np.where(sel != 0, alt_p, q)
    pick = _nonconverged(H, cfg, syn_all[112:160], 6, 1, probs=probs)
    jobs["cs16sel"] = (cfg, syn_all[112:160][pick], probs[pick], sel[pick], alt_p)
    for name, (cfg, syn, probs, sel, alt_p) in jobs.items():
        t0 = time.time()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            rs = list(ex.map(_one_shot, [(H, cfg, syn[b], None if probs is None else probs[b]) for b in range(len(syn))]))
        r = {k: np.stack([x[k] for x in rs]) for k in ("osdw", "osd0", "bp", "converged", "iters")}
        print(name, "%.1fs" % (time.time() - t0), "converged", r["converged"], "weights", r["osdw"].sum(1), r["osd0"].sum(1), flush=True)
        extra = {} if sel is None else dict(prior_select=pack(sel), alt_prob=alt_p)
        np.savez_compressed(os.path.join(HERE, f"l29k_golden_{name}.npz"), cfg=repr(cfg), code_sha256=digest,
                            syn=pack(syn), osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]),
                            converged=r["converged"], iters=r["iters"], **extra)


def main():
    H = l29k().hz
    digest = hashlib.sha256(H.indptr.tobytes() + H.indices.tobytes()).hexdigest()
    if len(sys.argv) > 1 and sys.argv[1] == "e15":
        return make_e15(H, digest)
    if len(sys.argv) > 1 and sys.argv[1] == "r4":
        return make_round4(H, digest)
    q = 0.06
    rng = np.random.default_rng(29524)
    err = (rng.random((4, H.shape[1])) < q).astype(np.uint8)
    syn = np.ascontiguousarray((H.astype(np.int32) @ err.T.astype(np.int32) % 2).T.astype(np.uint8))
    # one easy shot so that the BP-converged branch is in the fixture as well
    easy = (rng.random((1, H.shape[1])) < 0.01).astype(np.uint8)
    syn = np.concatenate([syn, (H.astype(np.int32) @ easy.T.astype(np.int32) % 2).T.astype(np.uint8)])
    for name, nshots, cfg in (
        ("e10", 5, dict(error_rate=q, max_iter=10, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_e", osd_order=10)),
        ("cs3", 2, dict(error_rate=q, max_iter=10, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=3)),
    ):
        t0 = time.time()
        sub = np.ascontiguousarray(syn[-nshots:])
        r = OracleDecoder(H, **cfg).decode_batch(sub)
        print(name, "%.1fs" % (time.time() - t0), "converged", r["converged"], "weights", r["osdw"].sum(1), flush=True)
        np.savez_compressed(os.path.join(HERE, f"l29k_golden_{name}.npz"), cfg=repr(cfg), code_sha256=digest,
                            syn=pack(sub), osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]),
                            converged=r["converged"], iters=r["iters"])


if __name__ == "__main__":
    main()
