"""Stream of asynchronous packed host calls on the headline workload (what bench.py's host_to_host.stream_packed_all_outputs times),
small enough to trace:  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stream_trace -- python3 tools/host_stream_probe.py
then  python tools/host_stream_probe.py --parse gpurun_out/stream_trace  prints the kernel timeline."""
import glob, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
    rows.sort()
    t0 = rows[0][0]
    for a, b, k in rows[-60:]:
        print("%10.3f %10.3f %8.3f  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, k))
    sys.exit(0)
from bp_osd_amd import BpOsdDecoder
from bp_osd_amd.codes import h1922
H = h1922(compute_logicals=False).hz
m, n = H.shape
B = 131072; q = 0.05
nsl = int(os.environ.get("NSL", "3")); ncalls = int(os.environ.get("NCALLS", "12"))
rng = np.random.default_rng(0)
syn = np.empty((B, m), np.uint8)
for lo in range(0, B, 16384):
    e = (rng.random((16384, n)) < q).astype(np.int32)
    syn[lo:lo + 16384] = (np.asarray(H.astype(np.int32) @ e.T) % 2).T
dec = BpOsdDecoder(H, error_rate=q, max_iter=0, bp_method="ms", ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
wm, wn = (m + 63) // 64, (n + 63) // 64
p_syn = dec.pinned_empty((B, wm), np.uint64); p_syn[:] = dec.pack_rows(syn)
bufs = [dict(osdw=dec.pinned_empty((B, wn), np.uint64), osd0=dec.pinned_empty((B, wn), np.uint64), bp=dec.pinned_empty((B, wn), np.uint64),
             conv=dec.pinned_empty((B,)), iters=dec.pinned_empty((B,), np.int32)) for _ in range(nsl)]
issue = lambda b: dec.decode_batch_packed_into(p_syn, b["osdw"], b["osd0"], b["bp"], b["conv"], b["iters"], wait=False)
lanes = [issue(bufs[k]) for k in range(nsl)]
dec.synchronize()
th = time.perf_counter(); stamps = []
for k in range(ncalls):
    sl = k % nsl
    if k >= nsl:
        dec.synchronize(lanes[sl])
    stamps.append(time.perf_counter() - th)
    lanes[sl] = issue(bufs[sl])
dec.synchronize()
dt = (time.perf_counter() - th) / ncalls
print("slots", nsl, "ms per call %.2f" % (1e3 * dt), "issue times (ms)", [round(1e3 * s, 1) for s in stamps], flush=True)
