#!/bin/bash
# Texture-addresser / L1 counter passes for bench.py (is a kernel bound by the number of cache lines its gathers and scatters touch?)
# Usage (on the GPU box): bash tools/pmc_ta.sh TAG --config l29k_ms_e15
set -o pipefail
TAG=${1:-r04}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcta_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --cpu-sample 0 --host-steps 0 --no-pipeline $@"
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $REPO/bench.py $ARGS > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/g$i.log; }
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'bposd::' in k:
            acc[(k.replace('void bposd::', '')[:34], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(acc.items()):
        print(f"{k:36s} {c:36s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
rm -rf $OUT/g*/
