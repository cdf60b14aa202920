#!/bin/bash
# Round-5 evidence in one GPU call: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the headline command (pipelined and one
# step at a time) and of configs[4]; the phase clocks of every elimination of an L29k launch (-DBPOSD_OSD_DIAG build, BPOSD_LIB=...).
# Summaries land in gpurun_out/r05p/; the judged copies go to profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05p
mkdir -p $OUT
cd $REPO
export PYTHONPATH=$REPO
bash tools/profile_bench.sh r05_pipe > $OUT/profile_pipe.log 2>&1; echo "profile default done"
bash tools/profile_bench.sh r05_nopipe --no-pipeline > $OUT/profile_nopipe.log 2>&1; echo "profile nopipe done"
bash tools/profile_bench.sh r05_l29k --config l29k_ms_e15 > $OUT/profile_l29k.log 2>&1; echo "profile l29k done"
cd $REPO
for t in pipe nopipe l29k; do
  f=$(find $REPO/gpurun_out/prof_r05_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$t.csv
  python tools/pmc_traffic_summary.py $REPO/gpurun_out/prof_r05_$t > $OUT/pmc_traffic_$t.json 2>/dev/null
done
rm -rf $REPO/gpurun_out/prof_r05_* $REPO/gpurun_out/pmc_r05_*
if [ -n "$DIAG_LIB" ]; then
  BPOSD_LIB=$REPO/$DIAG_LIB BPOSD_OSD_DEBUG=1 timeout -k 10 600 python tools/l29k_slow_probe.py 1024 > $OUT/l29k_phase_clocks_all_eliminations.txt 2>&1; echo "phase clocks done"
fi
ls $OUT
