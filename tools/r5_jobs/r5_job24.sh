cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j24
export VPU_LIB_DIAG=1
for a in "--model vith --batch 12 --steps 6 --warmup 2" "--batch 8" "--model vitl --batch 8 --steps 6 --warmup 2" "--model vith --batch 8 --steps 6 --warmup 2"; do for m in "VPU_GEMM_K2_RBMIN=7" "VPU_GEMM_K2_RBMIN=6" "VPU_GEMM_K2_RBMIN=5" "VPU_GEMM_K2_RBMIN=7" "VPU_GEMM_K2_RBMIN=6" "VPU_GEMM_K2_RBMIN=5"; do echo "== $a $m"; env $m python3 bench.py $a --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j24/ab.txt
