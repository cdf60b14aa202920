#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats, then separate PMC passes for HBM traffic.
# Usage (from the repo root on the GPU box): bash tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --host-steps 0 --no-extras $@"   # (--no-extras: the headline configuration alone; the default line's other configurations launch the same kernel names at other batch sizes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace_bench.log 2>&1 || { echo trace failed; tail -5 $OUT/trace_bench.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch_bench.log 2>&1 || { echo pmc fetch failed; tail -5 $OUT/pmc_fetch_bench.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write_bench.log 2>&1 || { echo pmc write failed; tail -5 $OUT/pmc_write_bench.log; exit 1; }
find $OUT -name "*.csv" | head -20
