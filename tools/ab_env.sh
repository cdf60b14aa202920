#!/bin/bash
# Same-box A/B of environment settings on the headline bench line (steps one at a time), alternating.
# Usage: REPS=3 bash tools/ab_env.sh "BPOSD_LAYOUT_ITERS=400000" "BPOSD_LAYOUT_ITERS=1000000"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in $(seq 1 ${REPS:-3}); do
  for setting in "$@"; do
    env $setting timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --host-steps 0 ${ARGS:---no-pipeline} > /tmp/ab.json 2>/tmp/ab.err || { echo "$setting FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$setting" $rep <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("rep", sys.argv[2], "%-36s"%sys.argv[1], "value %.4g"%d["value"], "bp_ms %.2f"%d["kernel_ms"]["bp"], "isolated %.2f"%d["kernel_ms_isolated"]["bp"], "model", d.get("roofline_lds",{}).get("bit_pass_bank_model"))
PY
  done
done
