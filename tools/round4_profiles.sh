#!/bin/bash
# Round-4 evidence in one GPU call: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the default command (pipelined and
# one step at a time), of configs[4] and of the reference's [[900,36,10]] workload; SQ counter passes of the default kernel and of
# product-sum (vector instructions per edge-iteration after the two-division restatement); the 2-rank rehearsal of the N > 1
# path on one GPU.  Summaries land in gpurun_out/r04p/; the judged copies go to profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04p
mkdir -p $OUT
cd $REPO
export PYTHONPATH=$REPO
bash tools/profile_bench.sh r04_pipe > $OUT/profile_pipe.log 2>&1; echo "profile default done"
bash tools/profile_bench.sh r04_nopipe --no-pipeline > $OUT/profile_nopipe.log 2>&1; echo "profile nopipe done"
bash tools/profile_bench.sh r04_l29k --config l29k_ms_e15 > $OUT/profile_l29k.log 2>&1; echo "profile l29k done"
bash tools/profile_bench.sh r04_hgp900 --config hgp900_ms_cs42 > $OUT/profile_hgp900.log 2>&1; echo "profile hgp900 done"
bash tools/pmc_sq.sh r04_default > $OUT/sq_counters_bp_local_kernel.txt 2>&1; echo "sq default done"
bash tools/pmc_sq.sh r04_ps_clip20 --config h1922_ps_cs60_clip20 > $OUT/sq_counters_h1922_ps_cs60_clip20.txt 2>&1; echo "sq ps clip done"
bash tools/pmc_sq.sh r04_ps_noclip --config h1922_ps_cs60 > $OUT/sq_counters_h1922_ps_cs60.txt 2>&1; echo "sq ps noclip done"
cd $REPO
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 \
    --cpu-sample 0 --host-steps 0 --rehearse-on-one-gpu > $OUT/bench_rehearsal_2ranks.out 2> $OUT/bench_rehearsal_2ranks.err; echo "rehearsal rc $?"
grep '^{' $OUT/bench_rehearsal_2ranks.out > $OUT/bench_rehearsal_2ranks.json   # (gloo prints its connection lines on stdout; RCCL runs do not)
for t in pipe nopipe l29k hgp900; do
  f=$(find $REPO/gpurun_out/prof_r04_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$t.csv
  python tools/pmc_traffic_summary.py $REPO/gpurun_out/prof_r04_$t > $OUT/pmc_traffic_$t.json 2>/dev/null
done
rm -rf $REPO/gpurun_out/prof_r04_* $REPO/gpurun_out/pmc_r04_*
ls $OUT
