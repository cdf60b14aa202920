#!/bin/bash
# configs[4] (l29k_ms_e15) with 2 / 3 / 4 lanes (HIP streams with their own workspaces) and as many steps in flight, alternating.
# Round 5: 11.63-11.68 k / 11.49-11.68 k / 11.63 k syndromes/s over 8 timed steps -- two lanes already pack the GPU (88 ms per step against
# 35 ms of BP + ~50 ms of CU time in eliminations).
for rep in 1 2; do
for cfg in "2 2" "3 3" "4 4"; do set -- $cfg
BPOSD_LARGE_LANES=$1 timeout -k 10 300 python bench.py --config l29k_ms_e15 --steps 8 --warmup 2 --slots $2 --cpu-sample 0 --host-steps 0 > /tmp/ab.json 2>/tmp/ab.err || { echo "lanes $1 FAILED"; tail -3 /tmp/ab.err; continue; }
python - $1 <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("lanes", sys.argv[1], "value %.0f"%d["value"], "ms_per_step %.1f"%d["ms_per_step"], "kernel_ms", {k: round(v,1) for k,v in d["kernel_ms"].items()}, "xcheck", d.get("cross_kernel_check",{}).get("identical"), flush=True)
PY
done; done
