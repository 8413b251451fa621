set -x
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/base_bench.json 2> gpurun_out/base_bench.err && \
bash tools/run_trace.sh r04_base_trace
