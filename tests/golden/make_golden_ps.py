"""Golden vectors for BASELINE configs[2] itself: H1922, product-sum BP, max_iter = n = 1922, osd_cs order 60, q = 0.05.

Same provenance as make_golden.py (this repository's oracle; no reference code).  Four fixtures, 2048 seeded shots each
(the `_pm` pair is the same two runs with ps_math = 1, i.e. tanh / log from bp_osd_amd/csrc/portable_math.h -- the
routines the GPU kernels evaluate -- instead of the platform libm; those two the GPU must reproduce bit for bit):

  ps_cs60_noclip.npz   ps_clip = 0  -- the reference formula (no clipping).  tanh rounds to 1
                                       within ~8 iterations, log((1+x)/(1-x)) = inf, inf - inf = NaN: every shot that
                                       has not converged by then ends with an all-NaN LLR vector and OSD runs in index
                                       order.  Per shot the fixture records the first non-finite iteration and whether
                                       the final LLRs hold inf / NaN.
  ps_cs60_clip20.npz   ps_clip = 20 -- the build-owned switch: check->bit messages clamped to [-20, 20]; no inf / NaN.

Per shot: packed error / osdw / osd0 / bp, converged, iters, first_nonfinite_iter, final_has_inf, final_has_nan, and -- for
shots whose final LLRs are finite -- the smallest gap between two distinct LLR values and the smallest |LLR| (the two
margins an ulp-level difference of tanh / log between libm and the device library would have to cross to change the
OSD column order or a hard decision).

    python tests/golden/make_golden_ps.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from bp_osd_amd.codes import h1922  # noqa: E402
from oracle import OracleDecoder  # noqa: E402

B = 2048
Q = 0.05
SEED = 2060


def pack(a):
    return np.packbits(np.asarray(a, dtype=np.uint8), axis=1)


def margins(llr):
    """(min gap between distinct sorted values, min |value|) per row; NaN for rows with non-finite entries."""
    gap = np.full(len(llr), np.nan)
    mabs = np.full(len(llr), np.nan)
    for b, row in enumerate(llr):
        if not np.isfinite(row).all():
            continue
        d = np.diff(np.sort(row))
        d = d[d > 0]
        gap[b] = d.min() if len(d) else np.inf
        mabs[b] = np.abs(row).min()
    return gap, mabs


def main():
    H = h1922(compute_logicals=False).hz
    rng = np.random.default_rng(SEED)
    err = (rng.random((B, H.shape[1])) < Q).astype(np.uint8)
    syn = np.ascontiguousarray((H @ err.T % 2).T.astype(np.uint8))
    only = sys.argv[1:]
    for name, clip, pm in (("noclip", 0.0, 0), ("clip20", 20.0, 0), ("noclip_pm", 0.0, 1), ("clip20_pm", 20.0, 1)):
        if only and name not in only:
            continue
        cfg = dict(error_rate=Q, max_iter=0, bp_method="ps", osd_method="osd_cs", osd_order=60, ps_clip=clip)
        t0 = time.time()
        r = OracleDecoder(H, ps_math=pm, **cfg).decode_batch(syn, want_diag=True)
        gap, mabs = margins(r["llr"])
        print(name, "%.0fs" % (time.time() - t0), "converged %.4f" % r["converged"].mean(), "mean iters %.1f" % r["iters"].mean(),
              "non-finite shots", int((r["first_nonfinite_iter"] > 0).sum()), "final NaN", int(r["final_has_nan"].sum()),
              "final inf", int(r["final_has_inf"].sum()), flush=True)
        np.savez_compressed(os.path.join(HERE, f"ps_cs60_{name}.npz"), cfg=repr(cfg), seed=SEED, q=Q, err=pack(err),
                            osdw=pack(r["osdw"]), osd0=pack(r["osd0"]), bp=pack(r["bp"]), converged=r["converged"],
                            iters=r["iters"], first_nonfinite_iter=r["first_nonfinite_iter"],
                            final_has_inf=r["final_has_inf"], final_has_nan=r["final_has_nan"],
                            min_gap=gap, min_abs=mabs, ps_math=pm,
                            # wrapping sum of the LLR bit patterns of every shot without NaN (NaN payloads are not portable)
                            llr_checksum=np.where(np.isnan(r["llr"]).any(axis=1), np.uint64(0),
                                                  np.ascontiguousarray(r["llr"]).view(np.uint64).sum(axis=1, dtype=np.uint64)))


if __name__ == "__main__":
    main()
