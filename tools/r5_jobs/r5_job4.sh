# round-5 job 4: full GPU suite + bench + attention micro-benchmarks
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j4
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/j4/pytest.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/j4/pytest.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/j4/bench.json 2> gpurun_out/j4/bench.err; echo "bench rc $?"
cut -c1-330 gpurun_out/j4/bench.json
for m in 1 2; do echo "== VPU_ATTN_ONEPASS=$m"; VPU_ATTN_ONEPASS=$m timeout -k 10 120 python3 tools/op_bench.py attn_bwd_window; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/j4/op_bench.txt
