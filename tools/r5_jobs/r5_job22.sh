cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j22
timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -k "k2_tile_height or k2_exact" > gpurun_out/j22/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/j22/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
for a in "--batch 4" "--batch 8" "--batch 12" "--model vith --batch 12 --steps 6 --warmup 2" "--model vitl --batch 8 --steps 6 --warmup 2"; do echo "== $a"; python3 bench.py $a --no-cpu-baseline 2>/dev/null | cut -c60-175; done | tee gpurun_out/j22/bench.txt
