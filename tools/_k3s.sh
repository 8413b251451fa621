set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k3 or colsum_batched" > gpurun_out/k3s_test.log 2>&1; echo "test rc $?" >> gpurun_out/k3s_test.log
tail -5 gpurun_out/k3s_test.log
grep -q " passed" gpurun_out/k3s_test.log || exit 1
GEMM_BENCH_K2=2,k3s GEMM_BENCH_ONLY="proj,fpn lin,head conv,head fuse,fpn ct,img out,img kproj 384,img kproj dgrad" timeout -k 10 300 python3 tools/gemm_bench.py 20 > gpurun_out/k3s_bench.log 2>&1
cat gpurun_out/k3s_bench.log | grep -v amdgpu
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e2e4_$name.json 2> gpurun_out/e2e4_$name.err || { tail -5 gpurun_out/e2e4_$name.err; return 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/e2e4_$name.json'))
r=d['roofline']
print('$name', d['value'], d['ms_per_step'], d['config']['final_loss'], r['kernel'], r['frac'], r['launches_per_step'], r['avg_launch_us'])
"
}
run old VPU_GEMM_K3=0 VPU_WGRAD_UNIFY=0 || exit 1
run k4p VPU_GEMM_K3=24 || exit 1
run k4p_k3s VPU_GEMM_K3=28 || exit 1
run k4p_k3s_ln3 VPU_GEMM_K3=28 VPU_LN_BWD_WGS=3 || exit 1
run k4p_ln3 VPU_GEMM_K3=24 VPU_LN_BWD_WGS=3 || exit 1
