set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04p
bash tools/profile_bench.sh r04_l29k --config l29k_ms_e15 > gpurun_out/r04p/profile_l29k.log 2>&1; echo "profile l29k done"
f=$(find gpurun_out/prof_r04_l29k -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04p/kernel_stats_l29k.csv
python tools/pmc_traffic_summary.py gpurun_out/prof_r04_l29k > gpurun_out/r04p/pmc_traffic_l29k.json
rm -rf gpurun_out/prof_r04_l29k
cat gpurun_out/r04p/pmc_traffic_l29k.json; grep "bposd::" gpurun_out/r04p/kernel_stats_l29k.csv | cut -c1-160
