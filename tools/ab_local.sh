# A/B of the BP kernels on the headline workload (serial steps, isolated kernel times):
# LDS kernel shape 2, local-edge kernel (2 checks/thread at <= 80 / <= 64 VGPRs, 1 check/thread)
for v in ${VARIANTS:-2 16 17 18 19 20 21}; do
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --host-steps 0 --no-pipeline --variant $v > /tmp/ab.json 2>/tmp/ab.err || { echo "v$v FAILED"; tail -3 /tmp/ab.err; continue; }
  python - $v <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
it=d["bp_iterations_mean"]*d["config"]["per_gpu_batch"]
print("variant", sys.argv[1], "value %.0f"%d["value"], "bp_ms %.2f"%d["kernel_ms"]["bp"], "osd_ms %.2f"%d["kernel_ms"]["osd"], "mean iters %.2f"%d["bp_iterations_mean"], "LER", d["logical_error_rate"], "ns/syn-it %.3f"%(d["kernel_ms"]["bp"]*1e6/it))
PY
done
