cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j8
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -k "layernorm" > gpurun_out/j8/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/j8/pytest.log
for m in "VPU_LN_FWD_RW=1" "VPU_LN_FWD_RW=2"; do echo "== $m"; env $m timeout -k 10 120 python3 tools/op_bench.py layernorm_fwd; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/j8/ln.txt
for m in "VPU_LN_FWD_RW=1" "VPU_LN_FWD_RW=2" "VPU_LN_FWD_RW=1" "VPU_LN_FWD_RW=2"; do echo "== bench $m"; env $m python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-175; done | tee gpurun_out/j8/bench_ab.txt
