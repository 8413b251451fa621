cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j10
for k in 0 2 3 4 5 6 8 0; do echo "== bench VPU_GEMM_K2_SKEW_ALL=1 VPU_GEMM_K2_SKEW=$k"; VPU_GEMM_K2_SKEW_ALL=1 VPU_GEMM_K2_SKEW=$k python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
v = d['roofline']['all_gemm_variants']
print(d['value'], d['ms_per_step'], {k[19:]: x['ms'] for k, x in v.items() if 'k2_kernel' in k})"; done | tee gpurun_out/j10/skew_all.txt
