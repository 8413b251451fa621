cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j13
for b in 4 8; do for t in 160 120 64 40; do echo "== batch $b K2_MIN_TILES=$t"; VPU_GEMM_K2_MIN_TILES=$t python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j13/ab.txt
for t in 160 64; do echo "== vitl batch 8 K2_MIN_TILES=$t";  VPU_GEMM_K2_MIN_TILES=$t python3 bench.py --model vitl --batch 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c60-175; done | tee -a gpurun_out/j13/ab.txt
