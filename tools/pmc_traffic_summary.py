"""Per-kernel HBM traffic from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh.
usage: python tools/pmc_traffic_summary.py gpurun_out/prof_<tag>   -> mean KB per launch per kernel (raw counters), and
bytes per launch with the gfx950 correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE counts half the bytes of a
coalesced stream -> doubled; WRITE_SIZE exact)."""
import collections, csv, glob, json, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bposd::" in k:
            acc[k.replace("void bposd::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    f = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    w = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    out[k] = {"launches": len(c.get("FETCH_SIZE", [])), "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
print(json.dumps(out, indent=1))
