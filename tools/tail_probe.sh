#!/bin/bash
for mi in 0 600 300 150 60; do
  timeout -k 10 120 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --max-iter $mi > /tmp/tp.log 2>&1 || { echo "mi $mi FAILED"; tail -3 /tmp/tp.log; continue; }
  python - $mi <<'PY'
import json,sys
d=json.loads(open("/tmp/tp.log").read().strip().splitlines()[-1])
it=d['bp_iterations_mean']*d['config']['per_gpu_batch']
print(f"max_iter {sys.argv[1]:>5s}: bp {d['kernel_ms']['bp']:7.2f} ms  mean it {d['bp_iterations_mean']:6.2f}  ns/syn-iter {d['kernel_ms']['bp']*1e6/it:6.3f}  conv {d['bp_converged_fraction']:.4f} osd/step {d['osd_invocations_per_step']:.0f} osd ms {d['kernel_ms']['osd']:.2f}")
PY
done
