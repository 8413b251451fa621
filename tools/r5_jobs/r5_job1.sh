# round-5 job 1: lazy-zero tests at bench shapes, bench baseline, serialized kernel trace, attention micro-benchmarks
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j1
python3 -m pytest tests/test_model_gpu.py -x -q -k "lazy_zero" > gpurun_out/j1/pytest.log 2>&1; echo "pytest rc $?" 
tail -5 gpurun_out/j1/pytest.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/j1/bench.json 2> gpurun_out/j1/bench.err; echo "bench rc $?"
cut -c1-400 gpurun_out/j1/bench.json
python3 tools/op_bench.py attn layernorm > gpurun_out/j1/op_bench.txt 2>&1; cat gpurun_out/j1/op_bench.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/j1 -o ser -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/j1/ser.log 2>&1
f=$(ls gpurun_out/j1/*kernel_trace.csv | head -1)
python3 tools/trace_seq.py $f > gpurun_out/j1/seq.txt 2>&1
python3 tools/trace_by_grid.py $f > gpurun_out/j1/by_grid.txt 2>&1
wc -l gpurun_out/j1/seq.txt
rm -f $f   # (tens of MB)
