#!/bin/bash
# Same-box A/B of environment settings on BASELINE configs[4] (l29k_ms_e15), alternating.
# Usage: REPS=2 bash tools/ab_env_l29k.sh "BPOSD_LARGE_APPLY=0" "BPOSD_LARGE_APPLY=1" "BPOSD_LARGE_APPLY=2"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in $(seq 1 ${REPS:-2}); do
  for setting in "$@"; do
    env $setting timeout -k 10 300 python bench.py --config l29k_ms_e15 --steps ${STEPS:-4} --warmup 1 --cpu-sample 0 --host-steps 0 > /tmp/ab.json 2>/tmp/ab.err || { echo "$setting FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$setting" $rep <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("rep", sys.argv[2], "%-24s"%sys.argv[1], "value %.0f"%d["value"], "ms_per_step %.1f"%d["ms_per_step"], "kernel_ms", {k: round(v,1) for k,v in d["kernel_ms"].items()}, "isolated", {k: round(v,1) for k,v in d["kernel_ms_isolated"].items()}, "LER", d["logical_error_rate"], "xcheck", d.get("cross_kernel_check",{}).get("identical"), flush=True)
PY
  done
done
