#!/bin/bash
# Round 5's numbers of record: the default line (headline + every other BASELINE configuration under "configs"), then one bench line
# per configuration WITH its CPU legs (JSON files under gpurun_out/<tag>/).
# Usage: bash tools/round5_numbers.sh <tag> [part]   part A (default): everything but the large code's CPU legs; part B: l29k_ms_e15
# with its ~12 minutes of CPU oracle; part C: probes.
set -o pipefail
TAG=${1:-r05}
PART=${2:-A}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONPATH=$REPO
if [ "$PART" = "A" ]; then
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_line.json 2> $OUT/bench.err && echo "bench default (with configs) done" &&
python bench.py --steps 5 --warmup 2 --no-pipeline --cpu-sample 0 --host-steps 0 --no-extras > $OUT/bench_h1922_ms_cs7_no_pipeline.json 2>> $OUT/bench.err &&
python bench.py --steps 5 --warmup 2 --p 0.0333 --cpu-sample 0 --host-steps 0 --no-extras > $OUT/bench_h1922_ms_cs7_q0333.json 2>> $OUT/bench.err &&
python bench.py --steps 3 --warmup 1 --config h1922_ms_osd0 --host-steps 0 > $OUT/bench_h1922_ms_osd0.json 2>> $OUT/bench.err && echo "osd0 done" &&
python bench.py --steps 3 --warmup 1 --config h1922_ps_cs60 --host-steps 0 > $OUT/bench_h1922_ps_cs60_noclip.json 2>> $OUT/bench.err && echo "ps noclip done" &&
python bench.py --steps 3 --warmup 1 --config h1922_ps_cs60_clip20 --host-steps 0 > $OUT/bench_h1922_ps_cs60_clip20.json 2>> $OUT/bench.err && echo "ps clip20 done" &&
python bench.py --steps 3 --warmup 1 --config h1922_ps_cs60_clip20 --ps-math-form 1 --host-steps 0 > $OUT/bench_h1922_ps_cs60_clip20_form1.json 2>> $OUT/bench.err && echo "ps clip20 form 1 done" &&
python bench.py --steps 3 --warmup 1 --config h1922_ps_cs60 --ps-math-form 1 --host-steps 0 --cpu-sample 0 > $OUT/bench_h1922_ps_cs60_noclip_form1.json 2>> $OUT/bench.err && echo "ps noclip form 1 done" &&
python bench.py --steps 10 --warmup 2 --config hgp400_ms_cs42 > $OUT/bench_hgp400_ms_cs42.json 2>> $OUT/bench.err && echo "hgp400 done" &&
python bench.py --steps 5 --warmup 2 --config hgp625_ms_cs42 --host-steps 0 > $OUT/bench_hgp625_ms_cs42.json 2>> $OUT/bench.err && echo "hgp625 done" &&
python bench.py --steps 5 --warmup 2 --config hgp900_ms_cs42 --host-steps 0 > $OUT/bench_hgp900_ms_cs42.json 2>> $OUT/bench.err && echo "hgp900 done" &&
python bench.py --steps 12 --warmup 2 --config l29k_ms_e15 --host-steps 0 --cpu-sample 0 > $OUT/bench_l29k_ms_e15_gpu_only.json 2>> $OUT/bench.err && echo "l29k (no CPU legs) done"
elif [ "$PART" = "B" ]; then
python bench.py --steps 12 --warmup 2 --config l29k_ms_e15 --host-steps 0 > $OUT/bench_l29k_ms_e15.json 2> $OUT/bench_l29k.err && echo "l29k done"
else
python tools/bp_iteration_cost.py 1 0 > $OUT/bp_iteration_cost.txt 2>&1 &&
python tools/latency_probe.py > $OUT/latency_probe.txt 2>&1 &&
python tools/osd_probe.py 2048 > $OUT/osd_probe.txt 2>&1 &&
echo "probes done"
fi
