#!/bin/bash
# SQ counter passes for bench.py (one pass per counter group; no tracing flags combined with --pmc)
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --cpu-sample 0 --host-steps 0 --no-pipeline $@"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
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
        print(f"{k:36s} {c:24s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
