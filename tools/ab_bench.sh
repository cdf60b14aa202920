#!/bin/bash
# A/B bench of library builds x BP shapes on the GPU box.  Usage: bash tools/ab_bench.sh lib1.so lib2.so ...
for lib in "$@"; do
  for v in 1 2 4; do
    BPOSD_LIB=$lib timeout -k 10 120 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --variant $v > /tmp/ab.log 2>&1 || { echo "$lib v$v FAILED"; tail -3 /tmp/ab.log; continue; }
    python - "$lib" $v <<'PY'
import json,sys
d=json.loads(open("/tmp/ab.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} v{sys.argv[2]} {d['value']/1e6:7.3f} M/s  bp {d['kernel_ms']['bp']:7.2f} ms osd {d['kernel_ms']['osd']:5.2f} ms  frac {d['roofline']['frac']:.2f}")
PY
  done
done
