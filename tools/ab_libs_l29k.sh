#!/bin/bash
# A/B of library builds on BASELINE configs[4] (l29k_ms_e15) IN ONE RUN.  Usage: REPS=2 bash tools/ab_libs_l29k.sh tools/_diag/libab_*.so
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    BPOSD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config l29k_ms_e15 --steps ${STEPS:-4} --warmup 1 --cpu-sample 0 --host-steps 0 > /tmp/ab.json 2>/tmp/ab.err || { echo "$lib FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$lib" $rep <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("rep", sys.argv[2], "%-24s"%sys.argv[1].split('/')[-1], "value %.0f"%d["value"], "ms_per_step %.1f"%d["ms_per_step"], "kernel_ms", {k: round(v,1) for k,v in d["kernel_ms"].items()}, "isolated", {k: round(v,1) for k,v in d["kernel_ms_isolated"].items()}, "LER", d["logical_error_rate"], flush=True)
PY
  done
done
