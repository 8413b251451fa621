set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k3_grouped or k2_grouped_wgrad" > gpurun_out/k3a_test.log 2>&1; echo "test rc $?" >> gpurun_out/k3a_test.log
tail -5 gpurun_out/k3a_test.log
grep -q "passed" gpurun_out/k3a_test.log && timeout -k 10 300 python3 tools/k3_bench.py 10 > gpurun_out/k3a_bench.log 2>&1
cat gpurun_out/k3a_bench.log
