#!/bin/bash
# configs[4] (l29k_ms_e15): capped bp_large_kernel grid x lanes (steps in flight).  Usage: CASES="256:2 128:3 128:4 160:3" bash tools/bp_grid_lanes_probe_l29k.sh
for rep in 1 2; do
for cs in ${CASES:-256:2 160:3 128:3 128:4 96:4}; do
g=${cs%%:*}; l=${cs##*:}
BPOSD_LARGE_BP_GRID=$g BPOSD_LARGE_LANES=$l timeout -k 10 300 python bench.py --config l29k_ms_e15 --steps 12 --warmup 3 --slots $l --cpu-sample 0 --host-steps 0 > /tmp/ab.json 2>/tmp/ab.err || { echo "grid $g lanes $l FAILED"; tail -3 /tmp/ab.err; continue; }
python - $g $l <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("bp grid", sys.argv[1], "lanes", sys.argv[2], "value %.0f"%d["value"], "ms_per_step %.1f"%d["ms_per_step"], "kernel_ms", {k: round(v,1) for k,v in d["kernel_ms"].items()}, "xcheck", d.get("cross_kernel_check",{}).get("identical"), flush=True)
PY
done; done
