cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j23
export VPU_LIB_DIAG=1
for a in "--batch 8" "--model vitl --batch 8 --steps 6 --warmup 2" "--model vith --batch 12 --steps 6 --warmup 2" "--batch 4"; do for m in "VPU_GEMM_K2_RB=7" "X=0" "VPU_GEMM_K2_RB=7" "X=0"; do echo "== $a $m"; env $m python3 bench.py $a --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j23/ab.txt
