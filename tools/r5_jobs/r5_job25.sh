cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j25
export VPU_LIB_DIAG=1
for a in "--batch 12 --steps 30 --warmup 5" "--batch 8" "--batch 4" "--model vitl --batch 8 --steps 6 --warmup 2" "--model vith --batch 12 --steps 6 --warmup 2"; do for m in 200 120 200 120; do echo "== $a NARROW_MIN=$m"; VPU_GEMM_K2_NARROW_MIN=$m python3 bench.py $a --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j25/ab.txt
