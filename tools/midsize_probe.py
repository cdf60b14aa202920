import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import hgp, circulant
from bench import make_batch
H = hgp(circulant(45, (0, 2, 5)), compute_logicals=False).hz
B = 32768
_, syn = make_batch(H, 0.05, B, seed=1)
d_syn = torch.from_numpy(syn).cuda()
out = torch.empty((B, H.shape[1]), dtype=torch.uint8, device='cuda')
d = BpOsdDecoder(H, error_rate=0.05, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
for _ in range(2):
    d.decode_batch_device(d_syn.data_ptr(), B, out.data_ptr()); d.synchronize()
t = d.last_timing()
print("hgp4050 (2025 x 4050): bp_ms %.1f osd_ms %.1f iters %d osd %d -> %.2f ns per syndrome-iteration, %.0f syndromes/s" % (
    t["bp_ms"], t["osd_ms"], t["bp_iterations"], t["osd_invocations"], t["bp_ms"] * 1e6 / t["bp_iterations"], B / ((t["bp_ms"] + t["osd_ms"]) * 1e-3)))
