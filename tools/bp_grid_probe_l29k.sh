#!/bin/bash
# configs[4] (l29k_ms_e15) with the persistent grid of bp_large_kernel capped (BPOSD_LARGE_BP_GRID): the kernel is bound by the memory
# system, not by the CUs it holds -- what does the step gain when it leaves CUs to the other lane's eliminations?
for rep in 1 2; do
for g in ${GRIDS:-256 192 160 128 96}; do
BPOSD_LARGE_BP_GRID=$g timeout -k 10 300 python bench.py --config l29k_ms_e15 --steps 8 --warmup 2 --cpu-sample 0 --host-steps 0 > /tmp/ab.json 2>/tmp/ab.err || { echo "grid $g FAILED"; tail -3 /tmp/ab.err; continue; }
python - $g <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("bp grid", sys.argv[1], "value %.0f"%d["value"], "ms_per_step %.1f"%d["ms_per_step"], "kernel_ms", {k: round(v,1) for k,v in d["kernel_ms"].items()}, "isolated", {k: round(v,1) for k,v in d["kernel_ms_isolated"].items()}, "xcheck", d.get("cross_kernel_check",{}).get("identical"), flush=True)
PY
done; done
