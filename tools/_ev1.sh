cd $GRAFT_REPO_ROOT
bash tools/run_stats.sh r04_final_stats > /dev/null 2>&1; echo "stats done"
bash tools/run_pmc_bench.sh r04_final_pmc > /dev/null 2>&1; python3 tools/pmc_traffic.py gpurun_out/r04_final_pmc gpurun_out/r04_final_pmc_traffic.json > gpurun_out/r04_final_pmc_traffic.txt 2>&1; head -8 gpurun_out/r04_final_pmc_traffic.txt
VPU_GEMM_K2_DIRECT=0 bash tools/run_pmc_gemm.sh r04_sq_k2_old > /dev/null 2>&1
VPU_GEMM_K2_DIRECT=1 bash tools/run_pmc_gemm.sh r04_sq_k2_new > /dev/null 2>&1
python3 tools/pmc_sq_summary.py gpurun_out/r04_sq_k2_old k2_kernel > gpurun_out/r04_sq_k2_old.txt 2>&1
python3 tools/pmc_sq_summary.py gpurun_out/r04_sq_k2_new k2_kernel > gpurun_out/r04_sq_k2_new.txt 2>&1
grep -c "" gpurun_out/r04_sq_k2_old.txt gpurun_out/r04_sq_k2_new.txt
