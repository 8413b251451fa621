set -x
cd $GRAFT_REPO_ROOT
for k3 in 0 1; do
VPU_GEMM_K3=$k3 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e2e_k3_$k3.json 2> gpurun_out/e2e_k3_$k3.err || exit 1
python3 -c "
import json
d=json.load(open('gpurun_out/e2e_k3_$k3.json'))
r=d['roofline']
print('K3=$k3', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['launches_per_step'], r['avg_launch_us'])
for k,v in r['all_gemm_variants'].items():
    if 'grouped' in k: print('   ',k,v)
"
done
