cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j15
for b in 12; do for m in 0 3 1 0 3 1; do echo "== batch $b VPU_NECK_LANES=$m"; VPU_NECK_LANES=$m python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j15/ab.txt
