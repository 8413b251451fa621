cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j11
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -s -k "split_attention" > gpurun_out/j11/pytest.log 2>&1; echo "pytest rc $?"; grep "split attention\|passed\|failed\|Error" gpurun_out/j11/pytest.log | tail -30
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -x -q -k "tiny or vitb_forward or vitb_bf16 or bitwise or bench_shape_bf16" > gpurun_out/j11/pytest_model.log 2>&1; echo "pytest model rc $?"; tail -3 gpurun_out/j11/pytest_model.log
for m in "VPU_XATTN_SPLIT=1" "VPU_XATTN_SPLIT=4" "VPU_XATTN_SPLIT=2" "VPU_XATTN_SPLIT=1" "VPU_XATTN_SPLIT=4"; do echo "== bench $m"; env $m python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-175; done | tee gpurun_out/j11/bench_ab.txt
