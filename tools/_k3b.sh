set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k3_" > gpurun_out/k3b_test.log 2>&1; echo "test rc $?" >> gpurun_out/k3b_test.log
tail -5 gpurun_out/k3b_test.log
grep -q " passed" gpurun_out/k3b_test.log && GEMM_BENCH_K2=2,k3 GEMM_BENCH_ONLY="qkv fwd,fc1 fwd,fc2 fwd,fc2 dgrad,fc1 dgrad,qkv dgrad,square" timeout -k 10 300 python3 tools/gemm_bench.py 20 > gpurun_out/k3b_bench.log 2>&1
cat gpurun_out/k3b_bench.log
