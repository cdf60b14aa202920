#!/bin/bash
# A/B of library builds on the headline workload IN ONE RUN (boxes differ by a few per cent, so only numbers taken on the
# same box compare).  Usage: VARIANTS="22 24" REPS=2 bash tools/ab_libs.sh bp_osd_amd/libab_*.so
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    for v in ${VARIANTS:-22 24}; do
      BPOSD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --host-steps 0 --no-pipeline --variant $v > /tmp/ab.json 2>/tmp/ab.err || { echo "$lib v$v FAILED"; tail -3 /tmp/ab.err; continue; }
      python - "$lib" $v $rep <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
it=d["bp_iterations_mean"]*d["config"]["per_gpu_batch"]
print("rep", sys.argv[3], "%-28s"%sys.argv[1].split('/')[-1], "variant", sys.argv[2], "bp_ms %.2f"%d["kernel_ms"]["bp"], "isolated %.2f"%d["kernel_ms_isolated"]["bp"], "ns/syn-it %.3f"%(d["kernel_ms"]["bp"]*1e6/it), "LER", d["logical_error_rate"])
PY
    done
  done
done
