set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k3_" > gpurun_out/k3c_test.log 2>&1; echo "test rc $?" >> gpurun_out/k3c_test.log
tail -5 gpurun_out/k3c_test.log
grep -q " passed" gpurun_out/k3c_test.log || exit 1
export GEMM_BENCH_ONLY="qkv fwd,fc1 fwd,fc2 dgrad"
echo "== prime, no stagger" > gpurun_out/k3c_bench.log
GEMM_BENCH_K2=2,k3 timeout -k 10 200 python3 tools/gemm_bench.py 20 >> gpurun_out/k3c_bench.log 2>&1
for st in 4 8 12; do
echo "== stagger $st" >> gpurun_out/k3c_bench.log
VPU_GEMM_K3_STAGGER=$st GEMM_BENCH_K2=k3 timeout -k 10 200 python3 tools/gemm_bench.py 20 >> gpurun_out/k3c_bench.log 2>&1
done
echo "== NOEPI" >> gpurun_out/k3c_bench.log
VPU_GEMM_NOEPI=1 GEMM_BENCH_K2=1,2,k3 timeout -k 10 200 python3 tools/gemm_bench.py 20 >> gpurun_out/k3c_bench.log 2>&1
cat gpurun_out/k3c_bench.log
