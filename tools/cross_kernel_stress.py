"""Cross-kernel stress at batch sizes the CPU oracle does not reach: OSD-bound syndromes (nearly all non-converged) decoded
by the tuned kernels three times each and compared, over all five outputs, with the same batch on a second implementation
(generic LDS kernel, or the any-degree kernel via bposd_set_bp_variant(h, 64)) -- the kind of run that exposes a rare race
(DESIGN.md 4.8).  usage: python tools/cross_kernel_stress.py   (output of record: profiles/r03_cross_kernel_stress.txt)"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922, hgp, circulant
for name, H in (("h1922", h1922(compute_logicals=False).hz), ("hgp4050", hgp(circulant(45, (0, 2, 5)), compute_logicals=False).hz)):
    m, n = H.shape; q = 0.085; B = 65536 if m < 1000 else 4096
    rng = np.random.default_rng(3)
    err = (rng.random((B, n)) < q).astype(np.uint8); syn = np.ascontiguousarray((np.asarray(H @ err.T) % 2).T.astype(np.uint8))
    kw = dict(error_rate=q, max_iter=12, bp_method="ms", ms_scaling_factor=0, osd_method="osd0")
    g = BpOsdDecoder(H, **kw); g.set_bp_variant(64 if m > 1000 else 1)
    want = dict(osdw=g.decode_batch(syn, want_osd0=True, want_bp=True).copy(), osd0=g.batch_osd0.copy(), bp=g.batch_bp.copy(), conv=g.batch_converge.copy(), iters=g.batch_iter.copy())
    print(name, "reference kernel", g.bp_kernel_info()["kernel"], "non-converged fraction %.3f" % (~want["conv"]).mean(), flush=True)
    for rep in range(3):
        d = BpOsdDecoder(H, **kw)
        got = dict(osdw=d.decode_batch(syn, want_osd0=True, want_bp=True), osd0=d.batch_osd0, bp=d.batch_bp, conv=d.batch_converge, iters=d.batch_iter)
        bad = {k: int((got[k] != want[k]).reshape(B, -1).any(axis=1).sum()) for k in want}
        print(name, d.bp_kernel_info()["kernel"], "rep", rep, "mismatching shots", bad, flush=True)
import scipy.sparse as sp
rng = np.random.default_rng(11)
big = hgp(circulant(62, (0, 2, 5)), compute_logicals=False).hz
irr = np.zeros((700, 1500), dtype=np.uint8)
for c in range(700):
    irr[c, rng.choice(1500, size=int(rng.integers(3, 9)), replace=False)] = 1
irr = sp.csr_matrix(irr[:, np.asarray(irr.sum(axis=0)).ravel() <= 8])
for name, H, B in (("3844x7688", big, 1024), ("irregular 700x%d" % irr.shape[1], irr, 32768)):
    m, n = H.shape; q = 0.05
    err = (rng.random((B, n)) < q).astype(np.uint8); syn = np.ascontiguousarray((np.asarray(H @ err.T) % 2).T.astype(np.uint8))
    kw = dict(error_rate=q, max_iter=10, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_cs", osd_order=4)
    g = BpOsdDecoder(H, **kw); g.set_bp_variant(64)
    want = dict(osdw=g.decode_batch(syn, want_osd0=True, want_bp=True).copy(), osd0=g.batch_osd0.copy(), bp=g.batch_bp.copy(), conv=g.batch_converge.copy(), iters=g.batch_iter.copy())
    print(name, "reference kernel", g.bp_kernel_info()["kernel"], "non-converged fraction %.3f" % (~want["conv"]).mean(), flush=True)
    for rep in range(3):
        d = BpOsdDecoder(H, **kw)
        got = dict(osdw=d.decode_batch(syn, want_osd0=True, want_bp=True), osd0=d.batch_osd0, bp=d.batch_bp, conv=d.batch_converge, iters=d.batch_iter)
        bad = {k: int((got[k] != want[k]).reshape(B, -1).any(axis=1).sum()) for k in want}
        print(name, d.bp_kernel_info()["kernel"], d.last_osd_kernel(), "rep", rep, "mismatching shots", bad, flush=True)
