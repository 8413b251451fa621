set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k3_grouped" > gpurun_out/k4a_test.log 2>&1; echo "test rc $?" >> gpurun_out/k4a_test.log
tail -5 gpurun_out/k4a_test.log
grep -q " passed" gpurun_out/k4a_test.log || exit 1
echo "== warm" > gpurun_out/k4a_bench.log
timeout -k 10 300 python3 tools/k3_bench.py 12 >> gpurun_out/k4a_bench.log 2>&1
echo "== cold (6 operand sets)" >> gpurun_out/k4a_bench.log
K3_BENCH_SETS=6 timeout -k 10 300 python3 tools/k3_bench.py 12 >> gpurun_out/k4a_bench.log 2>&1
cat gpurun_out/k4a_bench.log
