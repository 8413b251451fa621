# round-5 job 2: new window backward + rasteriser + backbone_forward + lazy tests; attention micro-benchmarks per mode
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j2
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "one_pass or attention or flash or thick or pue or disk" > gpurun_out/j2/pytest_ops.log 2>&1; echo "pytest ops rc $?"
tail -4 gpurun_out/j2/pytest_ops.log
grep "one-pass 2" gpurun_out/j2/pytest_ops.log | head -40
for m in 2 1 0; do echo "== VPU_ATTN_ONEPASS=$m"; VPU_ATTN_ONEPASS=$m timeout -k 10 120 python3 tools/op_bench.py attn_bwd_window; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/j2/op_bench.txt
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -k "lazy_zero or backbone_forward or scribble or tiny or public_coord" > gpurun_out/j2/pytest_model.log 2>&1; echo "pytest model rc $?"
tail -4 gpurun_out/j2/pytest_model.log
