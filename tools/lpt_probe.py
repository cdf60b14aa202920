import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922
code = h1922(compute_logicals=False); H = code.hz
rng = np.random.default_rng(1); q = 0.05; B = 131072
err = (rng.random((B, 1922)) < q).astype(np.uint8)
syn = np.asarray((H @ err.T) % 2).T.astype(np.uint8)
dec = BpOsdDecoder(H, error_rate=q, max_iter=1922, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
dec.decode_batch(syn)
it = dec.batch_iter.astype(np.int64); sw = syn.sum(1)
print("iters mean", it.mean(), "max-iter shots", (it == 1922).sum(), "total iters", it.sum())
order = np.argsort(-sw, kind="stable")
rank = np.empty(B, int); rank[order] = np.arange(B)
strag = np.where(it >= 1000)[0]
print("stragglers (>=1000 its):", len(strag), " rank quantiles in weight order:", np.quantile(rank[strag] / B, [0.1, 0.25, 0.5, 0.75, 0.9]))
print("corr(sw, iters)", np.corrcoef(sw, it)[0, 1], " mean sw", sw.mean(), "sw of stragglers", sw[strag].mean())
# simulated makespan: P = 1024 slots (256 CUs x 4 WGs) processing in given order, time = iters
import heapq
def makespan(seq, P=1024):
    h = [0] * P; heapq.heapify(h)
    for s in seq:
        t = heapq.heappop(h); heapq.heappush(h, t + it[s] + 3)
    return max(h)
print("makespan index order", makespan(range(B)), " LPT-by-syndrome-weight", makespan(order), " ideal", (it.sum() + 3 * B) / 1024)
