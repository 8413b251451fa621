cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j16
for b in 4 8; do for m in "X=0" "VPU_GEMM_INLAUNCH=1" "VPU_GEMM_K3=60" "VPU_GEMM_RING=1" "VPU_GEMM_K3=60 VPU_GEMM_INLAUNCH=1" "X=0"; do echo "== batch $b $m"; env $m python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j16/ab.txt
