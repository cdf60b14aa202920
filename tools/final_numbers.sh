#!/bin/bash
# The round's numbers of record in one GPU call: bench lines per configuration (JSON files under gpurun_out/<tag>/),
# rocprofv3 kernel stats + PMC passes of the headline command, probes.  Usage: bash tools/final_numbers.sh <tag>
set -o pipefail
TAG=${1:-final}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONPATH=$REPO
python bench.py --steps 20 --warmup 5 > $OUT/bench_h1922_ms_cs7.json 2> $OUT/bench.err && echo "bench default done" &&
python bench.py --steps 5 --warmup 2 --no-pipeline --cpu-sample 0 --host-steps 0 > $OUT/bench_h1922_ms_cs7_no_pipeline.json 2>> $OUT/bench.err &&
python bench.py --steps 5 --warmup 2 --p 0.0333 --cpu-sample 0 --host-steps 0 > $OUT/bench_h1922_ms_cs7_q0333.json 2>> $OUT/bench.err &&
python bench.py --steps 3 --warmup 1 --config h1922_ms_osd0 --host-steps 0 > $OUT/bench_h1922_ms_osd0.json 2>> $OUT/bench.err &&
python bench.py --steps 3 --warmup 1 --config h1922_ps_cs60 --cpu-sample 0 --host-steps 0 > $OUT/bench_h1922_ps_cs60_noclip.json 2>> $OUT/bench.err &&
python bench.py --steps 3 --warmup 1 --config h1922_ps_cs60_clip20 --cpu-sample 0 --host-steps 0 > $OUT/bench_h1922_ps_cs60_clip20.json 2>> $OUT/bench.err &&
python bench.py --steps 10 --warmup 2 --config hgp400_ms_cs42 > $OUT/bench_hgp400_ms_cs42.json 2>> $OUT/bench.err &&
python bench.py --steps 4 --warmup 1 --config l29k_ms_e15 --cpu-sample 0 --host-steps 0 > $OUT/bench_l29k_ms_e15.json 2>> $OUT/bench.err &&
python tools/bp_iteration_cost.py 1 0 > $OUT/bp_iteration_cost.txt 2>&1 &&
BP_METHOD=ps python tools/bp_iteration_cost.py 1 0 > $OUT/bp_iteration_cost_ps.txt 2>&1 &&
echo "bench configs done" &&
python tools/latency_probe.py > $OUT/latency_probe.txt 2>&1 &&
python tools/osd_probe.py 2048 > $OUT/osd_probe.txt 2>&1 &&
python tools/latency_crossover.py 256 2048 8192 > $OUT/latency_crossover.txt 2>&1 &&
echo "probes done"
# (rocprofv3 kernel stats, SQ counter passes and the 2-rank rehearsal: tools/round3_profiles.sh)
