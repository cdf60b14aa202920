"""Random small codes through the one-wave-per-elimination OSD kernel (and whichever BP kernel applies) against the oracle:
irregular random matrices, rank-deficient ones (repeated rows), sizes on both sides of the kernel's shape boundaries
(m = 64 q, n + 1 = 64 w), every OSD method, both tie policies.  usage: python tools/stress_small_osd.py [ncases] [seed]"""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from oracle import OracleDecoder

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
shapes = [(6, 13), (40, 100), (63, 126), (64, 127), (65, 128), (128, 255), (129, 256), (190, 447), (192, 400), (193, 448), (300, 625), (320, 639),
          (321, 640), (432, 900), (448, 959), (449, 960), (100, 959), (448, 500)]
bad = 0
used = {}
for case in range(ncases):
    m, n = shapes[case % len(shapes)]
    H = np.zeros((m, n), np.uint8)
    rdeg = np.zeros(m, int)
    cap = max(3, min(14, (4 * n) // m + 2))
    for j in range(n):  # column degree 1 .. 4, row degrees capped (the kernels take check degree <= 16, bit degree <= 8)
        d = int(rng.integers(1, 5))
        free = np.flatnonzero(rdeg < cap)
        rows = rng.choice(free, size=min(d, len(free)), replace=False)
        H[rows, j] = 1
        rdeg[rows] += 1
    for i in range(m):  # no empty rows
        if not H[i].any():
            H[i, rng.integers(n)] = 1
    if case % 3 == 0 and m > 8:  # rank deficiency: a few repeated / combined rows
        H[1] = H[0]
        if (H[2] ^ H[3]).any() and (H[2] ^ H[3]).sum() <= 16: H[m - 1] = H[2] ^ H[3]
    H = sp.csr_matrix(H)
    rank = np.linalg.matrix_rank(H.toarray().astype(float)) if m * n < 30000 else None
    q = rng.uniform(0.03, 0.12)
    err = (rng.random((96, n)) < q).astype(np.uint8)
    syn = np.ascontiguousarray((H @ err.T % 2).T.astype(np.uint8))
    method, order = [("osd0", 0), ("osd_cs", 5), ("osd_cs", 30), ("osd_e", 6), ("osd_e", 11), ("osd_cs", 64)][case % 6]
    tie = case % 2
    kw = dict(error_rate=float(q), max_iter=int(rng.integers(1, 6)), bp_method="ms", ms_scaling_factor=0.7, osd_method=method, osd_order=order,
              sort_tie_policy=tie, osd_e_bit_order=(case // 2) % 2)
    try:
        g = BpOsdDecoder(H, **kw)
    except ValueError as e:  # order beyond n - rank
        kw["osd_order"] = 2
        g = BpOsdDecoder(H, **kw)
    o = OracleDecoder(H, **kw)
    g.set_osd_variant(2)  # the wave kernel where it applies (auto keeps small calls on the workgroup kernel)
    out = g.decode_batch(syn)
    ref = o.decode_batch(syn, want_llr=False)
    ok = (out == ref["osdw"]).all() and (g.batch_osd0 == ref["osd0"]).all() and (g.batch_iter == ref["iters"]).all() and (g.batch_bp == ref["bp"]).all()
    k = (g.bp_kernel_info()["kernel"], g.last_osd_kernel())
    used[k] = used.get(k, 0) + 1
    nonconv = int((~g.batch_converge).sum())
    if not ok:
        bad += 1
        print("MISMATCH", case, (m, n), kw, k, "non-converged", nonconv, flush=True)
print("cases", ncases, "mismatches", bad, "kernels used", used)
sys.exit(1 if bad else 0)
