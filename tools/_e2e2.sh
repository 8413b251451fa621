set -x
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e2e2_$name.json 2> gpurun_out/e2e2_$name.err || { tail -5 gpurun_out/e2e2_$name.err; return 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/e2e2_$name.json'))
r=d['roofline']
print('$name', d['value'], d['ms_per_step'], d['config']['final_loss'], r['kernel'], r['frac'], r['launches_per_step'], r['avg_launch_us'])
for k,v in r['all_gemm_variants'].items():
    if 'grouped' in k or '65536' in k: print('   ',k,v)
"
}
run k2_old VPU_GEMM_K3=0 VPU_WGRAD_UNIFY=0 || exit 1
run k3_unify VPU_GEMM_K3=1 VPU_WGRAD_UNIFY=1 || exit 1
run k4p_unify VPU_GEMM_K3=24 VPU_WGRAD_UNIFY=1 || exit 1
run k4p_nounify VPU_GEMM_K3=24 VPU_WGRAD_UNIFY=0 || exit 1
