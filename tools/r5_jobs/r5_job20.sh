cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j20
for b in 4; do for m in "X=0" "VPU_GEMM_SKINNY_M=4096" "VPU_GEMM_K2_MIN_TILES=70" "VPU_GEMM_SKINNY_M=4096 VPU_GEMM_K2_MIN_TILES=70" "X=0"; do echo "== batch $b $m"; env $m python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j20/ab.txt
