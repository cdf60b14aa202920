import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import hgp, regular_ldpc_seed
from bench import make_batch
H = hgp(regular_ldpc_seed(31, 31, 3, 3, seed=3), compute_logicals=False).hz
B = 65536
_, syn = make_batch(H, 0.04, B, seed=1)
d_syn = torch.from_numpy(syn).cuda()
out = torch.empty((B, H.shape[1]), dtype=torch.uint8, device='cuda')
for v in (2, 16):
    d = BpOsdDecoder(H, error_rate=0.04, max_iter=0, bp_method="ms", ms_scaling_factor=0.75, osd_method="osd_cs", osd_order=7)
    d.set_bp_variant(v)
    li = d.layout_info()
    for _ in range(3):
        d.decode_batch_device(d_syn.data_ptr(), B, out.data_ptr()); d.synchronize()
    t = d.last_timing()
    print("variant", v, "bp_ms %.2f osd_ms %.2f iters %d ns/syn-it %.3f" % (t["bp_ms"], t["osd_ms"], t["bp_iterations"], t["bp_ms"] * 1e6 / t["bp_iterations"]), "LDS-kernel layout", li)
