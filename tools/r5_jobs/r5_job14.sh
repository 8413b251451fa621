cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j14
timeout -k 10 500 python3 -m pytest tests/test_model_gpu.py -x -q -s -k "neck_lanes" > gpurun_out/j14/pytest_lanes.log 2>&1; rc=$?; echo "pytest lanes rc $rc"; grep "neck lanes\|passed\|failed\|Error\|error" gpurun_out/j14/pytest_lanes.log | tail -25
if [ $rc -ne 0 ]; then exit 1; fi
for b in 12 4; do for m in 0 1 0 1; do echo "== batch $b VPU_NECK_LANES=$m"; VPU_NECK_LANES=$m python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-175; done; done | tee gpurun_out/j14/ab.txt
