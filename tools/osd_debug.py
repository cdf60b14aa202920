"""Compare the OSD kernel's internal column bit-vectors with a host Gauss-Jordan (one syndrome)."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BPOSD_OSD_DEBUG"] = "1"; os.environ["BPOSD_OSD_DUMP"] = "/tmp/osd_dump.bin"
from bp_osd_amd import BpOsdDecoder
from oracle import OracleDecoder
from tests.golden_util import load
g = load("golden_h1922_osd_cs60.npz"); H = g["H"]; Hd = H.toarray().astype(np.uint8); syn = g["syn"]
kw = dict(g["cfg"]); b = 20
gpu = BpOsdDecoder(H, **kw); cpu = OracleDecoder(H, **kw)
out = gpu.decode_batch(syn[b:b + 1])
d = np.fromfile("/tmp/osd_dump.bin", dtype=np.uint64)
yvec = d[8:24]; tpos = d[24:88].astype(int); colvec = d[88:88 + 1024].reshape(64, 16)
r = cpu.decode(syn[b]); o = cpu.osd(syn[b], r["llr"]); order = o["order"]; piv = o["pivot_flag"].astype(bool)
T = np.nonzero(~piv)[0]
npm = d[88 + 1024:88 + 1024 + 32]
gpu_np = np.array([(int(npm[j >> 6]) >> (j & 63)) & 1 for j in range(1922)], dtype=bool)
diff = np.nonzero(gpu_np != ~piv)[0]
print("non-pivot flags differ at sorted positions", diff[:20], "count", len(diff), "host npiv", int(piv.sum()), "gpu npiv", int((~gpu_np).sum()))
mism = np.nonzero(tpos[:60] != T[:60])[0]
print("tpos first mismatch T-index", mism[:5], "gpu", tpos[mism[:5]], "host", T[mism[:5]])
print("tpos match first 60:", (tpos[:60] == T[:60]).all(), tpos[:8], T[:8])
# host GJ
M = np.concatenate([Hd[:, order], syn[b][:, None]], axis=1).copy(); m = M.shape[0]; used = np.zeros(m, bool)
for j in range(1922):
    if not piv[j]: continue
    p = np.nonzero(M[:, j] & ~used)[0][0]; used[p] = True
    rows = np.nonzero(M[:, j])[0]; rows = rows[rows != p]; M[rows] ^= M[p]
y = M[:, -1]
w0_host = int(y[used].sum()); w0_gpu = sum(bin(int(v)).count("1") for v in yvec)
print("w0 host", w0_host, "gpu", w0_gpu)
# column weights: popcount is row-permutation invariant
for a in (0, 1, 20, 51, 59):
    hw = int(M[used][:, T[a]].sum()); gw = sum(bin(int(v)).count("1") for v in colvec[a])
    hyx = int((y[used] ^ M[used][:, T[a]]).sum()); gyx = sum(bin(int(v ^ u)).count("1") for v, u in zip(colvec[a], yvec))
    print("T-index", a, "pos", T[a], "col weight host", hw, "gpu", gw, "| y^col host", hyx, "gpu", gyx)
hp = int((y[used] ^ M[used][:, T[20]] ^ M[used][:, T[51]]).sum()); gp = sum(bin(int(u ^ v ^ w)).count("1") for u, v, w in zip(yvec, colvec[20], colvec[51]))
print("pair (20,51): host", hp + 2, "gpu", gp + 2)
