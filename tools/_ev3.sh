cd $GRAFT_REPO_ROOT
bash tools/run_stats.sh r04_final_stats > /dev/null 2>&1; echo "stats done"
bash tools/run_pmc_bench.sh r04_final_pmc > /dev/null 2>&1; python3 tools/pmc_traffic.py gpurun_out/r04_final_pmc gpurun_out/r04_final_pmc_traffic.json > gpurun_out/r04_final_pmc_traffic.txt 2>&1; head -4 gpurun_out/r04_final_pmc_traffic.txt
python3 tools/k4_drift.py > gpurun_out/r04_k4_drift_dcs.txt 2>&1; K4_DRIFT_DCS=0 python3 tools/k4_drift.py > gpurun_out/r04_k4_drift_classic.txt 2>&1
