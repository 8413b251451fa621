set -x
cd $GRAFT_REPO_ROOT
for k3 in 1 24; do
echo "=== VPU_GEMM_K3=$k3" >> gpurun_out/wq_trace.log
VPU_GEMM_K3=$k3 timeout -k 10 200 python3 tools/wq_trace.py >> gpurun_out/wq_trace.log 2>&1
done
cat gpurun_out/wq_trace.log | grep -v amdgpu.ids
