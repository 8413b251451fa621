# round-5 job 5: A/B knobs on one box + launch count
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j5
for m in "VPU_ATTN_ONEPASS=1" "VPU_ATTN_ONEPASS=2" "VPU_ATTN_ONEPASS=2 VPU_ATTN_WINP_NU=2"; do echo "== $m"; env $m timeout -k 10 120 python3 tools/op_bench.py attn_bwd_window; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/j5/op_bench.txt
VPU_ATTN_ONEPASS=2 VPU_ATTN_WINP_NU=2 timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -k "one_pass or large_scores" > gpurun_out/j5/pytest_nu2.log 2>&1; echo "pytest nu2 rc $?"; tail -3 gpurun_out/j5/pytest_nu2.log
for c in 2048 4096 1024; do echo "== VPU_LN_FWD_CAP=$c"; VPU_LN_FWD_CAP=$c timeout -k 10 120 python3 tools/op_bench.py layernorm_fwd; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/j5/op_bench.txt
for w in 2 3 4; do echo "== VPU_LN_BWD_WGS=$w"; VPU_LN_BWD_WGS=$w timeout -k 10 120 python3 tools/op_bench.py layernorm_bwd; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/j5/op_bench.txt
for m in "VPU_ATTN_ONEPASS=1" "VPU_ATTN_ONEPASS=2 VPU_ATTN_WINP_NU=2" "VPU_ATTN_ONEPASS=1"; do echo "== bench $m"; env $m python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-200; done | tee gpurun_out/j5/bench_ab.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/j5 -o ser -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/j5/ser.log 2>&1
f=$(ls gpurun_out/j5/*kernel_trace.csv | head -1)
python3 tools/trace_seq.py $f > gpurun_out/j5/seq.txt 2>&1
python3 tools/trace_by_grid.py $f > gpurun_out/j5/by_grid.txt 2>&1
wc -l gpurun_out/j5/seq.txt
rm -f $f
