#!/bin/bash
# Round-3 evidence in one GPU call: SQ counter passes (default kernel, reference example workload, product-sum with and
# without clip), rocprofv3 kernel stats of the default command and of the reference example workload, the 2-rank
# rehearsal of the N > 1 path on one GPU.  Summaries land in gpurun_out/r03/; the judged copies go to profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
cd $REPO
export PYTHONPATH=$REPO
bash tools/pmc_sq.sh r03_default > $OUT/sq_counters_bp_local_kernel.txt 2>&1; echo "sq default done"
bash tools/pmc_sq.sh r03_hgp400 --config hgp400_ms_cs42 > $OUT/sq_counters_hgp400_ms_cs42.txt 2>&1; echo "sq hgp400 done"
bash tools/pmc_sq.sh r03_ps_clip20 --config h1922_ps_cs60_clip20 > $OUT/sq_counters_h1922_ps_cs60_clip20.txt 2>&1; echo "sq ps clip done"
bash tools/pmc_sq.sh r03_ps_noclip --config h1922_ps_cs60 > $OUT/sq_counters_h1922_ps_cs60.txt 2>&1; echo "sq ps noclip done"
bash tools/profile_bench.sh r03_pipe > $OUT/profile_pipe.log 2>&1; echo "profile default done"
bash tools/profile_bench.sh r03_nopipe --no-pipeline > $OUT/profile_nopipe.log 2>&1; echo "profile nopipe done"
bash tools/profile_bench.sh r03_hgp400 --config hgp400_ms_cs42 > $OUT/profile_hgp400.log 2>&1; echo "profile hgp400 done"
cd $REPO
# N > 1 rehearsal: the launcher starts the ranks before any GPU call; both ranks decode on device 0, gloo gather
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 \
    --rehearse-on-one-gpu > $OUT/bench_rehearsal_2ranks.json 2> $OUT/bench_rehearsal_2ranks.err; echo "rehearsal rc $?"
for f in $(find $REPO/gpurun_out/prof_r03_* -name "*kernel_stats.csv"); do cp $f $OUT/$(echo $f | sed 's#.*/prof_\(r03_[a-z0-9]*\)/.*#\1#')_kernel_stats.csv; done
ls $OUT
