set -x
cd $GRAFT_REPO_ROOT
echo "=== VPU_GEMM_K3=24" > gpurun_out/wq_trace2.log
VPU_GEMM_K3=24 timeout -k 10 200 python3 tools/wq_trace.py >> gpurun_out/wq_trace2.log 2>&1
grep -v amdgpu.ids gpurun_out/wq_trace2.log | cut -c1-250
VPU_GEMM_K3=24 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e2e3.json 2> gpurun_out/e2e3.err || tail -5 gpurun_out/e2e3.err
python3 -c "
import json
d=json.load(open('gpurun_out/e2e3.json'))
r=d['roofline']
print('k4p_unify2', d['value'], d['ms_per_step'], d['config']['final_loss'], r['kernel'], r['frac'], r['launches_per_step'], r['avg_launch_us'])
"
