#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) of a bench command under two environment settings.
# Usage (GPU box): bash tools/pmc_traffic_ab.sh "BPOSD_LARGE_BP_RECORDS=0" "BPOSD_LARGE_BP_RECORDS=1" -- --config l29k_ms_e15
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SETTINGS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETTINGS+=("$1"); shift; done; shift
ARGS="--steps 1 --warmup 1 --cpu-sample 0 --host-steps 0 --no-pipeline $@"
cd /tmp && export TMPDIR=/tmp
for s in "${SETTINGS[@]}"; do
  OUT=$REPO/gpurun_out/pmcab_$(echo $s | tr -c 'A-Za-z0-9\n' '_')
  mkdir -p $OUT
  export $s
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "$s $c ..." 
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $REPO/bench.py $ARGS > $OUT/$c.log 2>&1 || { echo "$s $c failed"; tail -3 $OUT/$c.log; }
  done
  echo "== $s"; python3 $REPO/tools/pmc_traffic_summary.py $OUT
  rm -rf $OUT/pmc_*
done
