"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

Provenance: NO reference code is involved -- the reference's decode arithmetic lives in the absent
third-party `ldpc` package, so these vectors are produced by this repository's own
oracle (oracle/bposd_oracle.c), which is pinned by the README known-answer and cross-checked against
tests/ref_numpy.py.  They freeze the oracle's behaviour (regression) and are what the GPU path is
compared with on the GPU box, where neither /root/reference nor a slow oracle run is wanted.

    python tests/golden/make_golden.py
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from bp_osd_amd.codes import surface13, h1922, hgp  # noqa: E402
from oracle import OracleDecoder  # noqa: E402


def pack(a):
    return np.packbits(np.asarray(a, dtype=np.uint8), axis=1)


def run(H, syn, **kw):
    dec = OracleDecoder(H, **kw)
    r = dec.decode_batch(syn)
    return r


def main():
    out = {}
    # ---- S13: every syndrome, several decoder configurations
    s13 = surface13()
    syn = np.array(list(itertools.product([0, 1], repeat=6)), dtype=np.uint8)
    cfgs = [
        dict(error_rate=0.05, max_iter=13, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=7),
        dict(error_rate=0.05, max_iter=3, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=7),
        dict(error_rate=0.05, max_iter=3, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_e", osd_order=7),
        dict(error_rate=0.05, max_iter=2, bp_method="ps", ms_scaling_factor=0.0, osd_method="osd0", osd_order=0),
    ]
    for k, cfg in enumerate(cfgs):
        r = run(s13.hz, syn, **cfg)
        np.savez_compressed(os.path.join(HERE, f"golden_s13_cfg{k}.npz"), cfg=repr(cfg), syn=syn,
                            osdw=r["osdw"], osd0=r["osd0"], bp=r["bp"], converged=r["converged"],
                            iters=r["iters"], llr=r["llr"])
    # ---- [[400,16,6]] from the reference's seed matrix
    seed = np.loadtxt(os.path.join(HERE, "mkmn_16_4_6.txt")).astype(np.uint8)
    c400 = hgp(seed)
    rng = np.random.default_rng(400)
    q = 0.07
    err = (rng.random((96, 400)) < q).astype(np.uint8)
    syn = (c400.hx @ err.T % 2).T.astype(np.uint8)
    cfg = dict(error_rate=q, max_iter=8, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=10)
    r = run(c400.hx, syn, **cfg)
    np.savez_compressed(os.path.join(HERE, "golden_hgp400.npz"), cfg=repr(cfg), syn=pack(syn), err=pack(err),
                        osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]), converged=r["converged"],
                        iters=r["iters"], llr=r["llr"][:24])
    # ---- H1922, the benchmark code.  (a) README-style settings at p=0.05 (BP mostly converges),
    #      (b) elevated noise + short BP to drive every shot through OSD.
    c = h1922(compute_logicals=False)
    rng = np.random.default_rng(1922)
    q = 0.05
    err = (rng.random((96, 1922)) < q).astype(np.uint8)
    syn = (c.hz @ err.T % 2).T.astype(np.uint8)
    cfg = dict(error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=7)
    r = run(c.hz, syn, **cfg)
    np.savez_compressed(os.path.join(HERE, "golden_h1922_p05.npz"), cfg=repr(cfg), syn=pack(syn), err=pack(err),
                        osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]), converged=r["converged"],
                        iters=r["iters"], llr=r["llr"][:8])
    q = 0.08
    err = (rng.random((48, 1922)) < q).astype(np.uint8)
    syn = (c.hz @ err.T % 2).T.astype(np.uint8)
    for name, cfg in (
        ("cs7", dict(error_rate=q, max_iter=30, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_cs", osd_order=7)),
        ("cs60", dict(error_rate=q, max_iter=30, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=60)),
        ("e10", dict(error_rate=q, max_iter=30, bp_method="ms", ms_scaling_factor=0.0, osd_method="osd_e", osd_order=10)),
    ):
        r = run(c.hz, syn, **cfg)
        np.savez_compressed(os.path.join(HERE, f"golden_h1922_osd_{name}.npz"), cfg=repr(cfg), syn=pack(syn),
                            err=pack(err), osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]),
                            converged=r["converged"], iters=r["iters"], llr=r["llr"][:8])
        print(name, "converged", int(r["converged"].sum()), "of", len(syn))


if __name__ == "__main__":
    main()
