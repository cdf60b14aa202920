#!/bin/bash
# Same-box A/B of this tree against an earlier round's tree kept under _ab/<name>/ (its own bench.py, package and built library):
# the headline bench line, steps one at a time, alternating.  Usage: REPS=3 bash tools/ab_rounds.sh r03
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${REPS:-3}); do
  for tree in "$REPO" "$REPO/_ab/$1"; do
    (cd $tree && timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --host-steps 0 ${ARGS:---no-pipeline} > /tmp/ab.json 2>/tmp/ab.err) || { echo "$tree FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$tree" $rep <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("rep", sys.argv[2], "%-40s"%sys.argv[1][-24:], "value %.4g"%d["value"], "ms_per_step %.2f"%d["ms_per_step"], "bp_ms %.2f"%d["kernel_ms"]["bp"], "isolated %.2f"%d["kernel_ms_isolated"]["bp"], "osd %.2f"%d["kernel_ms_isolated"]["osd"], "LER", d["logical_error_rate"])
PY
  done
done
